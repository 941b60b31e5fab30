"""Device-side synthetic bands for the large benchmark configurations (torch is used for memory and RNG only).

Same recipe as `synthetic.synth_band` without the planted loops' host loop being a bottleneck:
Poisson(depth * (1 + k)^-alpha) per diagonal, 3x3 enrichments, weights 1/sqrt(marginal + 1) with NaN runs.
Returns torch tensors on `device`: raw f32 [n, ld], weight f64 [n], IR f64 [num], biases f64 [n].
"""
import numpy as np


def device_band(n, num, ld, mw, depth=60.0, alpha=1.0, nloops=200, seed=0, nan_frac=0.025, enrich=8.0, device='cuda',
                want_expected=True, structure=None):
    """structure: None, or keyword arguments of `synthetic.structure_fields` - TAD blocks, a compartment checkerboard and dense
    far-field patches multiply the rate (the same fields as the host generator's, applied slab by slab)."""
    import torch
    from . import synthetic
    fields = None
    if structure is not None:
        fields = {}
        for key, v in synthetic.structure_fields(n, num, seed, **structure).items():
            fields[key] = v.tolist() if key == 'patches' else torch.from_numpy(np.ascontiguousarray(v)).to(device)
        if 'tad_id' in fields:
            fields['tad_id'] = fields['tad_id'].long()
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    k = torch.arange(num, device=device, dtype=torch.float32)
    lam = depth * (1.0 + k) ** (-alpha)
    raw = torch.zeros((n, ld), dtype=torch.float32, device=device)
    rows = 8192
    rng = np.random.default_rng(seed)
    loops = []
    for _ in range(nloops):
        d = int(rng.integers(10, max(11, num - 15)))
        r = int(rng.integers(2, max(3, n - d - 2)))
        loops.append((r, d))
    loops.sort()
    for r0 in range(0, n, rows):
        r1 = min(n, r0 + rows)
        rate = lam.unsqueeze(0).expand(r1 - r0, num).clone()
        for (r, d) in loops:
            if r0 - 1 <= r <= r1:
                for dr in (-1, 0, 1):
                    for dc in (-1, 0, 1):
                        rr, kk = r + dr, d + dc - dr
                        if r0 <= rr < r1 and 0 <= kk < num:
                            rate[rr - r0, kk] *= enrich
        if fields is not None:
            rr_ = torch.arange(r0, r1, device=device).unsqueeze(1).expand(r1 - r0, num)
            cc_ = (rr_ + torch.arange(num, device=device).unsqueeze(0)).clamp(max=n - 1)
            rate = rate * synthetic.structure_gain(fields, rr_, cc_, xp=torch)
        blk = torch.poisson(rate, generator=g)
        rr = torch.arange(r0, r1, device=device).unsqueeze(1)
        blk[(rr + torch.arange(num, device=device).unsqueeze(0)) >= n] = 0
        raw[r0:r1, :num] = blk
    # marginals of the symmetric matrix restricted to the band (counts are integers: the f64 sums are exact in any order)
    colsum = torch.zeros(n, dtype=torch.float64, device=device)
    rowsum = torch.zeros(n, dtype=torch.float64, device=device)
    kk = torch.arange(num, device=device)
    for r0 in range(0, n, rows):
        r1 = min(n, r0 + rows)
        blk = raw[r0:r1, :num].to(torch.float64)
        rowsum[r0:r1] = blk.sum(dim=1)
        cc = torch.arange(r0, r1, device=device).unsqueeze(1) + kk.unsqueeze(0)
        sel = (cc < n) & (kk.unsqueeze(0) >= 1)
        colsum.index_add_(0, cc[sel], blk[sel])
    weight = 1.0 / torch.sqrt(rowsum + colsum + 1.0)
    nbad = int(round(n * nan_frac))
    if nbad > 0:
        run = max(1, (2 * nbad) // 3)
        start = int(rng.integers(n // 3, max(n // 3 + 1, 2 * n // 3 - run)))
        weight[start:start + run] = float('nan')
        if nbad - run > 0:
            idx = torch.from_numpy(rng.choice(n, size=nbad - run, replace=False)).to(device)
            weight[idx] = float('nan')
    if not want_expected:        # the library derives IR and the biases on the device (hpk_band.IR = NULL)
        return raw, weight, None, None
    IR = expected_on_device(raw, weight, n, num, mw)
    ok = ~((weight == 0) | torch.isnan(weight))
    biases = torch.zeros_like(weight)
    biases[ok] = 1.0 / weight[ok]
    return raw, weight, IR, biases


def expected_on_device(raw, weight, n, num, mw, rows=8192):
    """IR[d] = mean over diagonal d of the balanced values (raw * w_r) * w_c, stored pixels in masked bins left out of both
    the sum and the count, zero-count pixels counted as 0 (scripts/pyHICCUPS:150-156), in row slabs."""
    import torch
    device = raw.device
    kk = torch.arange(num, device=device)
    tot = torch.zeros(num, dtype=torch.float64, device=device)
    cnt = torch.zeros(num, dtype=torch.float64, device=device)
    for r0 in range(0, n, rows):
        r1 = min(n, r0 + rows)
        c = raw[r0:r1, :num].to(torch.float64)
        cc = torch.arange(r0, r1, device=device).unsqueeze(1) + kk.unsqueeze(0)
        inside = cc < n
        diag = (c * weight[r0:r1].unsqueeze(1)) * weight[cc.clamp(max=n - 1)]
        good = inside & ~(torch.isnan(diag) & (c != 0))
        diag = torch.where(good & (c != 0), diag, torch.zeros_like(diag))
        tot += diag.sum(dim=0)
        cnt += good.sum(dim=0).to(torch.float64)
    IR = tot / cnt          # (a diagonal without a single unmasked pixel: 0 / 0 = NaN, as numpy's mean of nothing)
    IR[:mw] = 0.0
    if num > n:
        IR[n:] = 0.0
    return IR
