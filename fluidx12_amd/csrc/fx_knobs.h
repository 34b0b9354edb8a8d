// fx_knobs.h -- the measurement / A-B switches of the kernel launchers (which kernel serves a geometry, chunk sizes, tile orders;
// docs/LAB.md lists them with what each measured).  None changes a result.  They are process-wide values set through fx_set_knob
// (include/fluidx_hip.h); the product reads no environment variable for them -- the Python harness forwards FLUIDX_<NAME> variables
// at load time for the tools' convenience (fluidx12_amd/capi.py).  A launcher looks its slot up once and reads the value per call.
#pragma once
#include <stdlib.h>

namespace fx {
int knob_slot(const char* name);               // -1: no such knob
const char* knob_at(int slot);                 // the value string, or nullptr while unset
inline int knob_int(int slot, int dflt) { const char* v = knob_at(slot); return v && *v ? atoi(v) : dflt; }
}  // namespace fx

#define FX_KNOB(NAME) ([]() -> const char* { static const int slot_ = ::fx::knob_slot(NAME); return ::fx::knob_at(slot_); }())
#define FX_KNOB_INT(NAME, D) ::fx::knob_int([]() -> int { static const int slot_ = ::fx::knob_slot(NAME); return slot_; }(), (D))
