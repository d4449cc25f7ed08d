#!/usr/bin/env python3
"""Headline benchmark: sites/sec of the Slater -> MPS conversion (BASELINE.json metric).

A step = one full C -> MPS conversion (all L sites): per-cut block diagonalisation, enumeration,
overlap / Schur complement and all tensor-block determinants.  Metric as SURVEY.md 8(d) defines it: the
timed region starts with C in HOST memory and ends with every tensor block and Schmidt value in HOST memory.

N = 1   workload = BASELINE config 3 sizes on one GPU: L=1024, chi_max=512, random complex hopping, seed 0
        (the configuration the metric is quoted on; it fits one GPU).
        value          host -> host over the K timed steps; the 1.5 GB download of conversion k runs on the
                       copy stream under the kernels of conversion k+1 (two results in flight, all K of them
                       complete in host memory before the clock stops)
        value_device   C resident in HBM -> tensors resident in HBM (round 1's headline; no PCIe)
        value_host_sync  host -> host, one conversion at a time, nothing overlapped (latency view)
        value_device_2inflight  device-resident, two conversions in flight on two contexts (two launch streams)
N > 1   the SAME seed-0 chain, its sites sharded over the ranks (contiguous cost-balanced ranges, boundary
        cuts recomputed by both neighbours, decisions reduced over the ranks): host C on rank 0 -> RCCL
        broadcast -> every rank converts its range and writes it through its own PCIe link into shared
        page-locked host memory -> rank 0 assembles ONE MPS.  scaling = "strong".  Extra object "replicas":
        every rank converts a whole chain of its own (seed = rank), HBM -> HBM, aggregate rate (weak).

`python bench.py --gpus N` without a launcher environment starts the N ranks itself (fresh child
processes, before this process touches a GPU) and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's ranks are used.
TMF_BENCH_SAME_DEVICE=1: rehearsal on a one-GPU box (all ranks on device 0, gloo instead of RCCL).
TMF_DRY_ENGINE=1: no GPU at all, stand-in engine - exercises spawn / rendezvous / assembly only (tests).

Prints ONE JSON line on rank 0 (see the driver contract in the task description).
"""
import argparse
import gc
import hashlib
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix = vector peak (AMD spec; the micro-architecture guide has no fp64 MFMA row)
PCIE_PEAK_GBS = 63.0      # PCIe Gen5 x16 host link (MI355X_MICROARCH.md)
PMC_FILE = os.path.join(ROOT, "profiles", "r03", "pmc_traffic.json")
REF_SUMMARY = os.path.join(ROOT, "tests", "golden", "full", "cfg3_rand_L1024_s0_chi512.npz")


def shard_sites(L, oc, world):
    from temfpy_amd.multi_gpu import shard_sites as f
    return f(L, oc, world)


def source_hashes():
    """SHA-1 of the kernel sources: a PMC file measured on other kernels is refused, not quoted."""
    out = {}
    d = os.path.join(ROOT, "temfpy_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            out[f] = hashlib.sha1(open(os.path.join(d, f), "rb").read()).hexdigest()
    return out


def _cpu_info():
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {"cpu_model": model, "nproc": len(os.sched_getaffinity(0)), "numpy": np.__version__}
    try:
        from threadpoolctl import threadpool_info
        info["blas"] = sorted({f"{p.get('internal_api')} {p.get('version')} ({p.get('threading_layer')}, "
                               f"{p.get('num_threads')} threads)" for p in threadpool_info()})
    except Exception:
        pass
    return info


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([int(p.get("num_threads", 1)) for p in threadpool_info()] or [1])
    except Exception:
        return len(os.sched_getaffinity(0))


def oracle_sample(C, chi, L, oc, sites):
    """Oracle (NumPy restatement of the reference, `kind: port`) on a bounded sample of sites: the
    reference's per-site work (two cut decompositions of which the sweep amortises one, overlap, Schur
    complement, one LAPACK determinant per matrix element)."""
    from oracle import slater_oracle as orc

    trunc = orc.as_trunc({"chi_max": chi})
    t0 = time.perf_counter()
    S = {}
    for i in sites:
        if i >= oc:
            bra = orc.cut_vectors(C, i + 1, trunc, "R")
            ket = orc.cut_vectors(C, i, trunc, "R" if i > oc else "LR")
            orc.site_tensor(bra, ket, "right")
        else:
            bra = orc.cut_vectors(C, i, trunc, "L")
            ket = orc.cut_vectors(C, i + 1, trunc, "L" if i + 1 < oc else "LR")
            orc.site_tensor(bra, ket, "left")
        for c in (bra, ket):
            p = c.lam**2
            S[c.x] = -(p[p > 0] * np.log(p[p > 0])).sum()
    return len(sites) / (time.perf_counter() - t0), S, time.perf_counter() - t0


def cpu_baseline(C, chi, L, oc, n_sample):
    """All BLAS threads on `n_sample` sites, then one thread on a quarter of them (SURVEY 8d asks for both)."""
    sites = sorted(set(np.linspace(0, L - 1, n_sample).astype(int).tolist()))
    v_all, S, t_all = oracle_sample(C, chi, L, oc, sites)
    out = {"value": round(v_all, 3), "unit": "sites/s", "cores": _blas_threads(), "kind": "port",
           "sample": f"{len(sites)} of {L} sites evenly spaced along the chain, {t_all:.1f} s of oracle time "
                     f"(NumPy/OpenBLAS threads = all cores; the per-matrix determinants are single-threaded LAPACK calls "
                     f"as in the reference)"}
    try:
        from threadpoolctl import threadpool_limits
        sub = sites[:: 4]
        with threadpool_limits(limits=1):
            v1, _, t1 = oracle_sample(C, chi, L, oc, sub)
        out["threads_1"] = {"value": round(v1, 3), "cores": 1, "sample": f"{len(sub)} sites, {t1:.1f} s"}
    except Exception as exc:   # threadpoolctl missing: say so instead of inventing a number
        out["threads_1"] = {"value": None, "note": f"not measured: {exc}"}
    out.update(_cpu_info())
    return out, S


def self_spawn(a):
    """`--gpus N` without a launcher: start N fresh ranks (before any GPU call here), relay rank 0's line."""
    from temfpy_amd.multi_gpu import spawn_ranks

    procs = spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus, stdout=subprocess.PIPE, text=True)
    out0 = None
    rc = 0
    try:
        out0, _ = procs[0].communicate(timeout=a.timeout)
        for p in procs:
            p.wait(timeout=a.timeout)
    except subprocess.TimeoutExpired:
        rc = 124
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.kill()
            rc = rc or 1
        elif p.returncode != 0:
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            rc = rc or p.returncode
        if r != 0 and p.stdout is not None:
            p.stdout.close()
    if out0:
        sys.stdout.write("".join(ln + "\n" for ln in out0.splitlines() if ln.startswith("{")))
        sys.stdout.flush()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--L", type=int, default=1024)
    ap.add_argument("--chi", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=64, help="sites timed with the CPU oracle (0 = skip)")
    ap.add_argument("--timeout", type=float, default=1500.0, help="self-spawn mode: seconds before the ranks are killed")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(a)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (before HIP / RCCL load: dmabuf IPC between the ranks)
    dry = os.environ.get("TMF_DRY_ENGINE") == "1"
    same_dev = os.environ.get("TMF_BENCH_SAME_DEVICE") == "1"
    # The contract is ONE JSON line on stdout.  Libraries underneath write to the C-level stdout (gloo announces its
    # peers there), so file descriptor 1 points to stderr for the rest of the run and the line goes to a saved copy.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()

    import torch
    import torch.distributed as dist
    from temfpy_amd import multi_gpu

    world_env = int(os.environ.get("WORLD_SIZE", 1))
    if world_env > 1:
        rank, world, dev = multi_gpu.init_rank(local=0 if same_dev else None, gloo=same_dev, dry=dry)
        world = dist.get_world_size()          # what the process group actually has
    else:
        rank, world, dev = 0, 1, None if dry else "cuda:0"
        if not dry:
            torch.cuda.set_device(0)

    from tests_inputs import random_hopping
    from temfpy_amd import slater
    from temfpy_amd.schmidt_utils import to_stopping_condition

    L, chi = a.L, a.chi
    oc = L // 2
    trunc = to_stopping_condition({"chi_max": chi})
    eng = multi_gpu.make_engine(dev, dry)
    # H -> C is outside the metric
    C0, N = slater.correlation_matrix(random_hopping(L, 0))

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()

    def timed(step, finish=None, collect=None):
        """W untimed + exactly K timed steps between barrier + synchronize; max over ranks."""
        for _ in range(a.warmup):
            step()
        if finish:
            finish()
        # Python's cyclic garbage collector is kept out of the timed region: a generation-2 collection over the
        # interpreter's ~10^6 live objects takes ~30 ms (found with rocprofv3 as a host stall between two conversions)
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        last = None
        for _ in range(a.steps):
            last = step()
            if collect:
                collect()
        if finish:
            finish()
        barrier()
        dt_ = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group.ctl)
            dt_ = float(t.item())
        return dt_, last

    out = {"metric": "sites/sec (Slater->MPS, L=%d chi=%d fp64)" % (L, chi), "unit": "sites/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "vs_baseline": None,
           "dtype": "c128 (fp64 complex)", "data": "synthetic"}

    if world > 1:
        # ---------------------------------------------------------------- N > 1: one chain, sites sharded
        tag = os.environ.get("TMF_SHM_TAG") or f"tmf{os.getppid()}p{os.environ.get('MASTER_PORT', '0')}"
        group = multi_gpu.ShardGroup(eng, tag, device=dev)
        busy = []

        # Pipelined like the one-GPU headline: step k enqueues conversion k (broadcast of C, kernels, DMA of this rank's
        # tensors into its shared page-locked segment), then completes conversion k - 1 (wait for its tensors, exchange
        # of the segment names, assembly on rank 0).  K conversions begin and K complete inside the timed region.
        pending, last_mps = [None], [None]

        def complete():
            if pending[0] is not None:
                last_mps[0] = group.convert_end(pending[0])
                busy.append(group.last_busy_ms)
                pending[0] = None
            return last_mps[0]

        def step_sharded():
            h = group.convert_begin(C0 if rank == 0 else None, trunc, oc, L)
            complete()
            pending[0] = h
            return last_mps[0]

        dt, _ = timed(step_sharded, finish=complete)
        mps = last_mps[0]
        ranges = shard_sites(L, oc, world)
        replicas = None
        if not dry:
            Cr, _ = slater.correlation_matrix(random_hopping(L, rank))
            d_C = torch.from_numpy(np.ascontiguousarray(Cr).reshape(-1)).to(dev)
            eng.coord = None      # replicas are independent conversions: no cross-rank decisions
            dt_r, _ = timed(lambda: eng.run(d_C, trunc, oc, L, download=False, threads=group.host_threads), finish=sync)
            replicas = {"value": round(world * L / (dt_r / a.steps), 2), "unit": "sites/s", "scaling": "weak",
                        "ms_per_step": round(dt_r / a.steps * 1e3, 3),
                        "workload": f"{world} independent L={L} chains (seed = rank), one per rank, C in HBM -> tensors in HBM"}
        if rank == 0:
            ms = dt / a.steps * 1e3
            busy_k = np.array(busy[-a.steps:])            # (K, world) busy ms of the timed steps
            out.update(value=round(L / (dt / a.steps), 2), ms_per_step=round(ms, 3), scaling="strong",
                       config={"workload": f"ONE L={L} random complex hopping chain (seed 0) Slater->MPS, chi_max={chi}, "
                                           f"svd_min=1e-6; sites sharded over {world} ranks {ranges}; host C on rank 0 -> "
                                           f"RCCL broadcast -> per-rank PCIe download into shared page-locked host memory -> "
                                           f"one assembled MPS on rank 0; the download of conversion k overlaps conversion k+1",
                               "N_fermions": N, "backend": dist.get_backend(), "contexts_per_rank": len(group.engines),
                               "busy_ms_per_rank": [round(float(x), 2) for x in busy_k.mean(axis=0)]},
                       replicas=replicas)
            if os.path.exists(REF_SUMMARY) and L == 1024 and chi == 512 and not dry:
                ref = np.load(REF_SUMMARY)
                out["max_abs_dS_vs_reference"] = float(np.abs(mps.entanglement_entropy(all_bonds=True) - ref["S"]).max())
                out["sites_assembled"] = int(sum(mps.sites[i] is not None for i in range(0, L, 37)))
            emit(out)
        del mps
        barrier()
        dist.destroy_process_group()
        return

    # -------------------------------------------------------------------- N = 1
    if dry:
        raise SystemExit("TMF_DRY_ENGINE needs --gpus > 1 (it rehearses the multi-rank plumbing)")
    group = None
    det_ms, det_flops, det_n = {}, {}, {}
    gemm_ms, gemm_fl, gemm_n, det_all = [], [], [], []
    split_ms, split_fl, split_n = [], [], []

    # (1) headline: host C in -> host tensors out, downloads overlapped with the next conversion
    results = []

    def step_host():
        m = eng.run(C0, trunc, oc, L, download="async")
        results.append(m)
        if len(results) > 1:
            # Conversion k + 1 is enqueued (its kernels overlap the download of conversion k); now wait for the tensors
            # of conversion k.  Not deeper: a conversion (25 ms) is shorter than its download (27 ms), so with two
            # downloads allowed to queue up the copy engine falls behind until the runtime stops overlapping copies and
            # kernels altogether (measured with tools/async_timeline.py: periods of 27, 27, 117 ms instead of 26.8).
            results.pop(0).wait()
        return m

    def finish_host():
        while results:
            results.pop(0).wait()

    dt, mps = timed(step_host, finish=finish_host)
    out_bytes = int(mps.shards[0].arrays["out"].nbytes)
    S_hip = mps.entanglement_entropy(all_bonds=True)
    stage_ms = {k: round(v * 1e3, 1) for k, v in mps.timings.items()}

    # (2) host -> host without overlap (one conversion at a time)
    dt_sync, _ = timed(lambda: eng.run(C0, trunc, oc, L, download=True))

    # (3) device-resident rate + per-kernel events (roofline)
    d_C = torch.from_numpy(np.ascontiguousarray(C0).reshape(-1)).to(dev)
    eng.time_gemm = True

    def collect():
        ki = eng.kernel_info         # HIP events on the launch stream, read back by tmf_sweep_info_get
        kind = ("ppt", "reduced", "direct")[ki.det_kind] + (f"/{ki.det_order}" if ki.det_kind else "")
        det_ms.setdefault(kind, []).append(ki.det_ms)
        det_flops[kind], det_n[kind] = ki.det_flops, int(ki.n_det)
        det_all.append(ki.det_all_ms)
        gemm_ms.append(ki.gemm_ms)
        gemm_fl.append(ki.gemm_flops)
        gemm_n.append(int(ki.n_gemm_launches))
        split_ms.append(list(ki.gemm_split_ms)), split_fl.append(list(ki.gemm_split_flops))
        split_n.append(list(ki.gemm_split_launches))

    dt_dev, mps_dev = timed(lambda: eng.run(d_C, trunc, oc, L, download=False), finish=sync, collect=collect)
    eng.time_gemm = False

    # (4) device-resident with TWO conversions in flight: a second context (its own launch stream) driven by a second
    # host thread (the staged C ABI releases the GIL); the kernels of the two conversions overlap and fill the CUs that
    # the latency-bound kernels of a single stream leave idle.  Each step here is a PAIR of conversions.
    import threading
    eng2 = multi_gpu.make_engine(dev, False)

    def pair():
        th = threading.Thread(target=lambda: eng2.run(d_C, trunc, oc, L, download=False, threads=16))
        th.start()
        eng.run(d_C, trunc, oc, L, download=False, threads=16)
        th.join()

    dt_pair, _ = timed(pair, finish=sync)

    ms = dt / a.steps * 1e3
    pmc_k, pmc_note = {}, "no PMC file"
    if os.path.exists(PMC_FILE) and L == 1024 and chi == 512:
        pj = json.load(open(PMC_FILE))
        stale = [f for f, h in source_hashes().items() if pj.get("source_sha1", {}).get(f) != h]
        if stale:
            pmc_note = f"{os.path.relpath(PMC_FILE, ROOT)} was measured on other kernel sources ({', '.join(stale[:4])}): refused"
        else:
            pmc_k, pmc_note = pj["kernels"], os.path.relpath(PMC_FILE, ROOT)
    roof = None
    if gemm_ms:
        # Dominant kernel by GPU time: the 64-wide-tile MFMA GEMM `gemm_kernel<cd, OPA, 64>` - rotation / overlap
        # products `slater.py:1071` and the GEMM share of the block diagonalisation.  achieved = 8 M N K summed over
        # the launches of one conversion / their summed duration (HIP events on the launch stream).
        n_l = int(np.mean(gemm_n))
        g_ms, g_fl = float(np.mean(gemm_ms)), float(np.mean(gemm_fl))
        ach = g_fl / (g_ms * 1e-3) / 1e12
        # (kernel names as rocprofv3 prints them: the 3M / 4M switch is a fourth template argument since round 3)
        kk = [pmc_k.get("tmf::gemm_kernel<tmf::cd, %d, 64, false>" % o) or pmc_k.get("tmf::gemm_kernel<tmf::cd, %d, 64>" % o) for o in (0, 1)]
        traffic = None
        if all(kk):
            traffic = round(sum(k_["hbm_bytes_per_launch"] * k_["launches"] for k_ in kk) / sum(k_["launches"] for k_ in kk))
        roof = {"bound": "mfma", "kernel": "tmf::gemm_kernel<tmf::cd, OPA, TN> (v_mfma_f64_16x16x4_f64)",
                "achieved": round(ach, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP64_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": pmc_note,
                "launches_per_step": n_l, "avg_launch_ms": round(g_ms / max(n_l, 1), 4),
                "flops_per_launch": round(g_fl / max(n_l, 1)),
                # algorithmic flops (8 M N K per complex product); the kernel forms complex products by the
                # 3M scheme, i.e. issues 6 M N K of them to the MFMA pipe
                "mfma_flops_issued_per_algorithmic_flop": 0.75,
                "mfma_pipe_frac": round(0.75 * ach / FP64_PEAK_TFLOPS, 4)}
        # the same figures per kernel instantiation, as they appear in a rocprofv3 kernel statistic: achieved =
        # flops_per_step / (launches_per_step x the statistic's average duration of that name)
        s_ms, s_fl, s_n = (np.mean(np.array(x, float), axis=0) for x in (split_ms, split_fl, split_n))
        roof["by_kernel"] = {
            name: {"launches_per_step": int(round(s_n[i])), "ms_per_step": round(float(s_ms[i]), 4),
                   "flops_per_step": round(float(s_fl[i])),
                   "achieved": round(float(s_fl[i]) / max(float(s_ms[i]) * 1e-3, 1e-12) / 1e12, 3),
                   "frac": round(float(s_fl[i]) / max(float(s_ms[i]) * 1e-3, 1e-12) / 1e12 / FP64_PEAK_TFLOPS, 4)}
            for i, name in enumerate(("tmf::gemm_kernel<tmf::cd, 0, 64> (A B)", "tmf::gemm_kernel<tmf::cd, 1, 64> (A^H B)",
                                      "tmf::gemm_kernel<tmf::cd, *, 16>")) if s_n[i] > 0}
        # the step itself is bound by the host link: every conversion moves C up and the tensors down
        link_bytes = out_bytes + C0.nbytes
        roof["pcie"] = {"bound": "pcie", "bytes_per_step": link_bytes, "achieved": round(link_bytes / (ms * 1e-3) / 1e9, 2),
                        "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": round(link_bytes / (ms * 1e-3) / 1e9 / PCIE_PEAK_GBS, 4)}
    dom = max(det_ms, key=lambda c: np.mean(det_ms[c])) if det_ms else None
    if dom is not None and roof is not None:
        avg_ms = float(np.mean(det_ms[dom]))
        kname = ("tmf::ppt_det_kernel<tmf::cd>" if dom == "ppt"
                 else f"tmf::reduced_det_kernel<tmf::cd, {dom.split('/')[1]}>" if dom.startswith("reduced")
                 else f"tmf::det_kernel<tmf::cd, {dom.split('/')[1]}, G>")
        all_ms = float(np.mean(det_all))
        # The determinant stage (90 % of the reference's time, `slater.py:828-869`).  `reference_work` counts the
        # REFERENCE's algorithm (one (8/3) n^3 LU per minor); the pivoted-exchange kernel evaluates
        # order-d minors of one shared exchange instead and executes almost none of those flops.  The hardware-true
        # figure of this kernel is its HBM rate (output bound): traffic / avg_launch_ms.
        if dom == "ppt":      # the kernel is instantiated per mask width: take the instance the PMC pass saw
            seen = [k for k in pmc_k if k.startswith("tmf::ppt_det_kernel<tmf::cd")]
            kname = max(seen, key=lambda k: pmc_k[k].get("launches", 0)) if seen else kname
        hbm = pmc_k.get(kname, {}).get("hbm_bytes_per_launch")
        roof["determinant_kernel"] = {
            "kernel": kname, "avg_launch_ms": round(avg_ms, 3), "dets_per_launch": det_n[dom],
            "all_det_launches_ms": round(all_ms, 3),
            "reference_work": {"flops_per_launch": det_flops[dom],
                               "rate_TFLOPs": round(det_flops[dom] / (avg_ms * 1e-3) / 1e12, 2)},
            "hbm": None if hbm is None else {"bound": "hbm", "traffic": hbm, "achieved": round(hbm / (avg_ms * 1e-3) / 1e9, 1),
                                             "peak": 8000.0, "unit": "GB/s", "frac": round(hbm / (avg_ms * 1e-3) / 8e12, 4)}}
    out.update(value=round(L / (dt / a.steps), 2), ms_per_step=round(ms, 3), scaling="none",
               config={"workload": f"L={L} random complex hopping (range 3, seed 0) Slater->MPS, chi_max={chi}, svd_min=1e-6; "
                                   f"host C in -> host tensors + Schmidt values out ({out_bytes / 1e9:.2f} GB per conversion), "
                                   f"download of conversion k overlapped with conversion k+1",
                       "N_fermions": N, "stage_ms": stage_ms,
                       "range_finder": {"subspace_iterations": eng.range_iterations_used,
                                        "smallest_captured_sigma": eng.range_floor}},
               value_device=round(L / (dt_dev / a.steps), 2), ms_per_step_device=round(dt_dev / a.steps * 1e3, 3),
               value_host_sync=round(L / (dt_sync / a.steps), 2),
               value_device_2inflight=round(2 * L / (dt_pair / a.steps), 2), roofline=roof)
    if os.path.exists(REF_SUMMARY) and L == 1024 and chi == 512:
        ref = np.load(REF_SUMMARY)   # the reference's own NumPy core at this size (tests/golden/make_golden_summary.py)
        out["max_abs_dS_vs_reference"] = float(np.abs(S_hip - ref["S"]).max())
        out["reference_cpu_in_build_container"] = {"value": round(L / float(ref["reference_wall_s"]), 3), "unit": "sites/s",
                                                   "cores": int(ref["reference_cores"]),
                                                   "note": "reference NumPy core, whole conversion, build container (not this box)"}
    if a.cpu_sample > 0:
        cb, S_ref = cpu_baseline(C0, chi, L, oc, a.cpu_sample)
        out["cpu_baseline"] = cb
        out["max_abs_dS_vs_oracle"] = float(max(abs(S_hip[b] - s) for b, s in S_ref.items()))
    emit(out)


if __name__ == "__main__":
    main()
