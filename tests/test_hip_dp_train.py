"""-m gpu: data-parallel training of the background NeRF over a process group (neuman_hip/dp.py, bkg_trainer.BackgroundNeRFTrainer(data_parallel=True));
reference train.py:26-28 (nn.DataParallel around both nets: one optimiser step on the whole batch, the network work split by rays).

What a one-GPU box can run of it: two ranks SHARING the GPU with the gradients reduced over gloo, and the RCCL collectives on a group of one
rank.  The reduced gradient must equal the single-process gradient of the concatenated batch to summation order, 20 steps' loss curves must
coincide.  Unmeasured on more than one device."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def env_for(rank, world, port):
    env = dict(os.environ)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def last_json(text):
    return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])


def check(rep, world):
    print("[dp train]", {k: rep[k] for k in ("backend", "world", "first_iteration_gradient_dev", "final_weight_dev")},
          "curves", [round(x, 5) for x in rep["curve_full"][:3]], "...", [round(x, 5) for x in rep["curve_full"][-2:]], "|",
          [round(x, 5) for x in rep["curve_dp"][:3]], "...", [round(x, 5) for x in rep["curve_dp"][-2:]])
    assert not rep["dead"] and rep["ranks_hold_equal_weights"] and rep["checkpoint_written_by_rank0"]
    assert rep["checkpoint_first_key"][0].startswith("module.")
    assert rep["first_iteration_gradient_dev"] < (1e-6 if world == 1 else 5e-6), rep["first_iteration_gradient_dev"]
    for a, b in zip(rep["first_iteration_terms_full"], rep["first_iteration_terms_dp"]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a))
    assert rep["curve_full"][-1] < rep["curve_full"][0]
    for a, b in zip(rep["curve_full"], rep["curve_dp"]):
        assert abs(a - b) <= 2e-3 * max(abs(a), 1e-3), (rep["curve_full"], rep["curve_dp"])


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_share_the_gpu_gradients_over_gloo(world):
    """world 8 = the node's size: eight shards of every batch, eight normaliser contributions and dead-flag votes in the all_gather, one all_reduce of
    the flat gradient buffer -- the summed gradient equals the full-batch one"""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "helpers", "dist_train_check.py"), "gloo"], env=env_for(r, world, port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    check(last_json(outs[0][0]), world)


def test_rccl_collectives_on_a_group_of_one_rank():
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "dist_train_check.py"), "nccl"], env=env_for(0, 1, free_port()),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    check(last_json(r.stdout), 1)
