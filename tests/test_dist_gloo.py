"""world_size-2 CPU test (gloo) of bench.py's multi-rank plumbing: torch.distributed rendezvous on 127.0.0.1,
z-slab assignment, broadcast of the RCCL unique-id bytes, barriers, max-over-ranks reduction and the single JSON
line from rank 0.  `--dry-run` skips all GPU work (and says so in the JSON), so this is never a measurement.
The slab arithmetic itself is verified on the GPU by tests/test_gpu_slabs.py (loop-back transport)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["scaling"] == "weak" and d["unit"] == "voxel-updates/s" and d["higher_is_better"] is True
    assert d["config"]["parallelism"].startswith("z-slab x2")
    assert "cpu_baseline" not in d                        # rank 0 at N = 1 only
    assert d["multi_rank_parity"].startswith("not checked (dry run")    # the key every N > 1 line carries; a GPU run replays the steps as one domain


def test_slab_partition_covers_the_grid():
    sys.path.insert(0, ROOT)
    import bench
    for Z in (256, 250, 64):
        for n in (1, 2, 3, 4, 8):
            slabs = [bench.slab_for_rank(Z, r, n) for r in range(n)]
            assert slabs[0][0] == 0 and sum(s[1] for s in slabs) == Z
            for (a, na), (b, _) in zip(slabs, slabs[1:]):
                assert a + na == b
    assert bench.step_bytes_per_voxel(40, "fp32") == 580 and bench.step_bytes_per_voxel(40, "fp16") == 534


def test_bench_started_directly_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run (WORLD_SIZE unset) must not fall back to one rank: the process
    becomes the launcher of two CHILD rank processes (never an exec) and passes their result and exit code on"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # a world size that contradicts --gpus is an error, not a silent 1-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_baseline_config_flags():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.BASELINE_CONFIGS[4] == (512, 80, "fp32") and bench.BASELINE_CONFIGS[2] == (128, 40, "fp32")
    assert bench.baseline_config_label(512, 512, 512, 80, "fp32", 8) == "BASELINE configs[3]"
    assert "40 instead of 80" in bench.baseline_config_label(512, 512, 512, 40, "fp32", 8)
    assert bench.baseline_config_label(256, 256, 256, 40, "fp32", 1).startswith("BASELINE configs[2]")
    assert bench.workload_grid(512, 8, "strong") == ((512, 512, 512), 8)
    # the default weak table changes row length at N = 8; weak256 keeps 256-wide rows for every N
    assert bench.workload_grid(256, 8, "weak")[0] == (512, 512, 512) and bench.workload_grid(256, 8, "weak256")[0] == (256, 256, 2048)
    assert bench.workload_grid(256, 4, "weak") == bench.workload_grid(256, 4, "weak256") == ((256, 256, 1024), 22)
