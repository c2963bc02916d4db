"""-m gpu: the N > 1 code path with real kernels under a process group, as far as a one-GPU box can execute it
(the 1/2/4/8-GPU curve itself is the driver's to measure):

* the RCCL gather collective on an initialised "nccl" group of ONE rank (render_sharded with force_collective), device tensors;
* bench.py's exact timed step through that group (`--dist`);
* two processes sharing the GPU, each rendering its own interleaved tiles on the device, frame assembled through gloo --
  bit-identical to the unsharded frame.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def env_for(rank, world, port):
    e = dict(os.environ)
    e.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    return e


def last_json(text):
    return json.loads([ln for ln in text.strip().splitlines() if ln.startswith("{")][-1])


def test_rccl_gather_on_a_group_of_one_rank():
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "dist_frame_check.py"), "nccl"], env=env_for(0, 1, free_port()),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = last_json(r.stdout)
    print(out)
    assert out["bit_identical"] and out["finite"] and out["tiles"] == 24


def test_bench_step_through_an_initialised_nccl_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--dist", "--timed-only"],
                       env=env_for(0, 1, free_port()), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout)
    print({k: line[k] for k in ("value", "ms_per_step", "n_gpus")})
    assert line["n_gpus"] == 1 and line["value"] > 1e5 and line["roofline"]["launches"] == 1


def test_two_ranks_share_the_gpu_and_assemble_real_tiles_over_gloo():
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "helpers", "dist_frame_check.py"), "gloo"], env=env_for(r, 2, port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    out = last_json(outs[0][0])
    print(out)
    assert out["bit_identical"] and out["finite"] and out["world"] == 2


def test_bench_self_launch_paths_on_a_one_gpu_box():
    """`python bench.py --gpus 2` started by hand on this one-GPU box prints ONE JSON error line (exit code 2) instead of dying on an
    assertion; with as many GPUs as asked it re-executes itself under torch.distributed.run (the CPU suite runs that start-up with gloo,
    tests/test_parallel_gloo.py); the line of an N = 1 run through the process group carries the multi-GPU evidence fields."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus_visible"] == n - 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--dist", "--timed-only"],
                       env=env_for(0, 1, free_port()), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    mg = last_json(r.stdout)["multi_gpu"]
    print(mg)
    assert mg["rccl_world"] == 1 and mg["backend"] == "nccl" and mg["rays_per_rank"] == [640000] and mg["frame_assembly_ms_per_rank"][0] > 0
