"""Helpers for the GPU parity tests: device buffers come from torch (plumbing),
the hot path is always called through the C ABI via ldpc_amd."""
import numpy as np


def dev_u64(shape):
    import torch
    return torch.zeros(shape, dtype=torch.int64, device="cuda")


def dev_u32(shape):
    import torch
    return torch.zeros(shape, dtype=torch.int32, device="cuda")


def dev_u8(shape):
    import torch
    return torch.zeros(shape, dtype=torch.uint8, device="cuda")


def dev_f64(shape):
    import torch
    return torch.zeros(shape, dtype=torch.float64, device="cuda")


def to_dev_u64(arr):
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)
    return torch.from_numpy(a.copy()).to("cuda")


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


def host_u32(t):
    return t.cpu().numpy().view(np.uint32)


def oracle_code(code):
    """Oracle handle for the same parity-check matrix as an ldpc_amd.LinearCode."""
    from oracle import oracle as orc
    ptr, idx = code.csr()
    return orc.Code(code.len(), ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))


def random_sparse_rows(rng, n, m, max_deg):
    rows = []
    for _ in range(m):
        k = int(rng.integers(0, min(n, max_deg) + 1))
        rows.append(sorted(int(x) for x in rng.choice(n, size=k, replace=False)))
    return rows
