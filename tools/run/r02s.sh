#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
hipcc -O2 --offload-arch=gfx950 tools/micro/ldslds2.cpp -o /tmp/ldslds2 2>/dev/null && /tmp/ldslds2
