"""Small utilities with the reference's names (temfpy/utils.py)."""
import logging

import numpy as np


def HT(M: np.ndarray) -> np.ndarray:
    """Hermitian conjugate (utils.py:8-10)."""
    return M.T.conj()


def n_slice(x: slice) -> int:
    """Number of elements of a slice (utils.py:13-16)."""
    return (x.stop - x.start) // (x.step or 1)


def normalize_SV(lam: np.ndarray, logger: logging.Logger) -> np.ndarray:
    """utils.py:99-103."""
    norm = np.linalg.norm(lam)
    logger.info(f"Norm of Schmidt values: {norm}")
    return lam / norm


def _device_svd(mats, device, want_vectors):
    """Singular values (descending) and optionally U, V^H of small square-padded matrices through the C ABI
    (``tmf_jacobi_compact_batched`` with accumulated rotations + ``tmf_gemm_batched``): the ``numpy.linalg.svd``
    calls of utils.py:90 and pfaffian.py:435.  No CPU path."""
    import torch

    from . import _native as nat
    from .gutzwiller import _gemm_recs, _gemm_tiles

    if not torch.cuda.is_available():
        raise nat.NativeError("temfpy_amd.utils needs a HIP device; there is no CPU fallback")
    lib, dev = nat.load(), torch.device(device)
    cplx = any(np.iscomplexobj(m) for m in mats)
    dt, ndt = (nat.TMF_C128, np.complex128) if cplx else (nat.TMF_F64, np.float64)
    el = np.dtype(ndt).itemsize
    ps = [max(m.shape) for m in mats]
    off = np.concatenate(([0], np.cumsum([(p * p + 1) & ~1 for p in ps]))).astype(np.int64)
    host = np.zeros(int(off[-1]) + 2, ndt)
    for m, p, o in zip(mats, ps, off[:-1]):
        pad = np.zeros((p, p), ndt)
        pad[: m.shape[0], : m.shape[1]] = m
        host[o: o + p * p] = pad.reshape(-1, order="F")
    stream = torch.cuda.current_stream(dev).cuda_stream
    d_M, d_X = torch.from_numpy(host).to(dev), torch.from_numpy(host.copy()).to(dev)
    d_W, d_V, d_G = (torch.zeros_like(d_M) for _ in range(3))
    so = np.concatenate(([0], np.cumsum(ps)))
    d_s = torch.zeros(int(so[-1]) + 1, dtype=torch.float64, device=dev)
    jd = np.zeros(len(mats), nat.jacobi_desc)
    items = []
    for i, (p, o) in enumerate(zip(ps, off[:-1])):
        jd[i] = (d_X.data_ptr() + el * o, d_W.data_ptr() + el * o, d_V.data_ptr() + el * o, d_s.data_ptr() + 8 * so[i], 0,
                 1e-300, p, p, p, p)
        items.append((d_M.data_ptr() + el * o, d_V.data_ptr() + el * o, d_G.data_ptr() + el * o, p, p, p, p, p, p))
    t_j = torch.from_numpy(jd.view(np.uint8).reshape(-1).copy()).to(dev)
    d_sw = torch.zeros(len(mats), dtype=torch.int32, device=dev)
    nat.check(lib.tmf_jacobi_compact_batched(dt, t_j.data_ptr(), len(mats), int(max(ps)), d_sw.data_ptr(), stream),
              "tmf_jacobi_compact_batched")
    if want_vectors:     # U S = M V
        g = _gemm_recs(items)
        tiles, tn = _gemm_tiles(g)
        t_g, t_t = torch.from_numpy(g.view(np.uint8).reshape(-1).copy()).to(dev), torch.from_numpy(tiles.reshape(-1).copy()).to(dev)
        nat.check(lib.tmf_gemm_batched(dt, 0, 1.0, 0.0, t_g.data_ptr(), t_t.data_ptr(), len(tiles), tn, stream), "tmf_gemm_batched")
    torch.cuda.synchronize(dev)
    nat.check_jacobi_sweeps(d_sw.cpu().numpy(), "Jacobi SVD (numpy.linalg.svd in utils.block_svd)")
    h_s, h_V, h_G = d_s.cpu().numpy(), d_V.cpu().numpy(), d_G.cpu().numpy()
    out = []
    for i, (m, p, o) in enumerate(zip(mats, ps, off[:-1])):
        s = h_s[so[i]: so[i] + p]
        if not want_vectors:
            out.append(s[: min(m.shape)])
            continue
        V = h_V[o: o + p * p].reshape(p, p).T
        US = h_G[o: o + p * p].reshape(p, p).T
        k = min(m.shape)
        U = US[: m.shape[0], :k] / np.where(s[:k] > 0, s[:k], 1.0)[None, :]
        out.append((U, s[:k], V[: m.shape[1], :k].conj().T))
    return out


def block_svd(CLR: np.ndarray, vL: np.ndarray, vR: np.ndarray, e: np.ndarray, degeneracy_tol: float = 1e-12,
              overwrite: bool = True, *, device: str = "cuda:0") -> tuple[np.ndarray, np.ndarray]:
    """Completes a block singular-value decomposition (utils.py:19-96): inside every group of (approximately)
    equal ``e`` the block ``vL^H CLR vR`` is SVD'd and ``vL``, ``vR`` are rotated (in place unless ``overwrite`` is
    false) so that they SVD ``CLR``.  Stand-alone helper kept for users of the reference's utility: the small SVDs
    run on the GPU (:func:`_device_svd`), the thin products around them are NumPy like the reference's ``einsum``
    calls.  The converter does not call it: its centre-bond pairing runs on the device (``engine.py``, stage E)."""
    assert vL.shape[1] == vR.shape[1] == e.size, "Mismatched number of eigenvalues and eigenvectors"
    assert vL.shape[0] == CLR.shape[0], "Mismatched row dimension"
    assert vR.shape[0] == CLR.shape[1], "Mismatched column dimension"
    if e.size == 0:
        return vL, vR
    if not overwrite:
        vL, vR = vL.copy(), vR.copy()
    (split_ix,) = np.nonzero(np.abs(np.diff(e)) > degeneracy_tol)
    split_ix = np.concatenate(([0], split_ix + 1, [len(e)]))
    groups = [np.arange(a, b) for a, b in zip(split_ix[:-1], split_ix[1:])]
    blocks = [vL[:, g].conj().T @ CLR @ vR[:, g] for g in groups]
    for g, (U, _, Vh) in zip(groups, _device_svd(blocks, device, True)):
        vL[:, g] = vL[:, g] @ U
        vR[:, g] = vR[:, g] @ Vh.conj().T
    return vL, vR
