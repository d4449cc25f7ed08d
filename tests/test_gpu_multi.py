"""Multi-process path on real hardware, rehearsed on ONE GPU: the worker pool behind
`slater.C_to_MPS(..., devices=[...])` with both workers on cuda:0 (gloo carries C and the decisions, RCCL
refuses two ranks on one device), and `bench.py --gpus 2` in the same rehearsal mode.  The assembled MPS must
equal the single-process conversion bit for bit."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT",
                                                             "TMF_DRY_ENGINE")}
    env.update(kw)
    return env


def test_device_pool_two_workers_one_gpu_bitwise():
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        from tests_inputs import random_hopping
        from temfpy_amd import slater
        L = 96
        C, _ = slater.correlation_matrix(random_hopping(L, 3))
        # the pool first: its workers must exist before this process touches the GPU
        sharded = slater.C_to_MPS(C, {"chi_max": 48}, devices=["cuda:0", "cuda:0"], as_tenpy=False)
        assert len(sharded.shards) == 2 and len(sharded.timings["busy_ms_per_rank"]) == 2
        single = slater.C_to_MPS(C, {"chi_max": 48}, as_tenpy=False)
        for b in range(L + 1):
            assert np.array_equal(sharded.bonds[b].lam, single.bonds[b].lam)
            assert np.array_equal(sharded.bonds[b].masks, single.bonds[b].masks)
        for i in range(L):
            a, b = sharded.sites[i], single.sites[i]
            assert a.det_always == b.det_always and len(a.blocks) == len(b.blocks)
            for x, y in zip(a.blocks, b.blocks):
                assert x[:5] == y[:5] and np.array_equal(x[5], y[5])
        print("ok")
    """ % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_two_ranks_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--L", "128", "--chi", "64",
                        "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], env=_clean_env(TMF_BENCH_SAME_DEVICE="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["replicas"]["value"] > 0


def test_rccl_branch_runs_once_with_world_size_one():
    """The data path of a multi-GPU node on the ONE GPU of this box: process group on the "nccl" backend (librccl loads,
    a communicator is created), C goes to the device through the page-locked buffer and is RCCL-broadcast (world size 1,
    collectives not skipped: TMF_SHARD_FORCE_COLLECTIVES), the decisions go through the gloo control group next to it, the
    tensors through the shared-memory segment and the assembly.  Complex and real input; the result equals the
    single-process conversion bit for bit.  (A multi-GPU node runs exactly this code with world size 8.)"""
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, torch
        import torch.distributed as dist
        from tests_inputs import random_hopping
        from temfpy_amd import slater, multi_gpu
        from temfpy_amd.schmidt_utils import to_stopping_condition
        rank, world, dev = multi_gpu.init_rank()
        assert dist.get_backend() == "nccl" and world == 1
        group = multi_gpu.ShardGroup(multi_gpu.make_engine(dev), os.environ["TMF_SHM_TAG"], device=dev)
        assert group.data_nccl and group.collective and dist.get_backend(group.ctl) == "gloo"
        tr = to_stopping_condition({"chi_max": 48})
        for L, real in ((96, False), (64, True)):
            H = random_hopping(L, 3)
            C, _ = slater.correlation_matrix(H.real.copy() if real else H)
            h = group.convert_begin(C, tr)                 # pipelined form: asynchronous download, second conversion begun
            h2 = group.convert_begin(C, tr)
            sharded, again = group.convert_end(h), group.convert_end(h2)
            single = slater.C_to_MPS(C, {"chi_max": 48}, as_tenpy=False)
            for m in (sharded, again):
                assert m.L == L and len(m.shards) == 1
                for b in range(L + 1):
                    assert np.array_equal(m.bonds[b].lam, single.bonds[b].lam)
                    assert np.array_equal(m.bonds[b].masks, single.bonds[b].masks)
                for i in range(L):
                    a, b = m.sites[i], single.sites[i]
                    assert a.det_always == b.det_always and len(a.blocks) == len(b.blocks)
                    for x, y in zip(a.blocks, b.blocks):
                        assert x[:5] == y[:5] and np.array_equal(x[5], y[5])
            assert set(sharded.info["checks"]) == set(multi_gpu.CHECK_NAMES)
            del sharded, again, m, a, b, x, y
        import ctypes
        loaded = open("/proc/self/maps").read()
        assert "librccl" in loaded, "the nccl backend did not load librccl"
        dist.barrier()
        dist.destroy_process_group()
        print("ok")
    """ % (ROOT, os.path.join(ROOT, "tests")))
    env = _clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                     TMF_SHM_TAG=f"tmfnccl{os.getpid()}", TMF_SHARD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
