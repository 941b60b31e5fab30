"""Host-side post-processing of the significant pixels: anchor detection and greedy clustering.

Counterpart of hicpeaks/callers.py:593-728 (`find_anchors`, `_cluster_core`, `local_clustering`).  It
touches only the few thousand pixels that survive scoring, so it stays on the host,
but it decides which pixels are printed, hence it is part of the drop-in.  Same names, arguments and
return values as the reference.  DBSCAN with min_samples=2 on integer pixels is the connected
components of the "distance <= eps" graph (every non-isolated point is a core point), so it is done
here with a union-find instead of importing scikit-learn.
"""
import numpy as np


_WALK_ALL_CELLS = False         # tests: local_clustering tests every cell of every anchor rectangle, as the reference does


def find_anchors(pos, min_count=3, min_dis=20000, wlen=200000, res=10000):
    """callers.py:593-634.  Returns a set of (summit, left, right) bins."""
    from scipy.signal import find_peaks, peak_widths

    min_dis = max(min_dis // res, 1)
    wlen = min(wlen // res, 10)
    pos = np.asarray(pos, dtype=np.int64)
    lo = int(pos.min()) - 1                                 # one empty bin on both sides
    signal = np.bincount(pos - lo, minlength=int(pos.max()) - lo + 2)
    refidx = range(lo, lo + signal.size)
    summits = find_peaks(signal, height=min_count, distance=min_dis)[0]
    # (the widths of all summits in one call: every peak's is computed on its own, as the reference's call per peak does)
    wl, wr = peak_widths(signal, summits, rel_height=1, wlen=wlen)[2:4] if len(summits) else ((), ())
    by_height = sorted(((signal[i], i, k) for k, i in enumerate(summits)), reverse=True)

    anchors = set()
    owner = {}                                              # bin -> anchor covering it
    for _, i, k in by_height:
        lb = refidx[int(np.round(wl[k]))]
        rb = refidx[int(np.round(wr[k]))]
        summit = refidx[i]
        if anchors:
            for b in range(lb, rb + 1):
                if b in owner:                              # overlaps an earlier (higher) anchor: merge
                    old = owner[b]
                    lb, rb, summit = min(lb, old[1]), max(rb, old[2]), old[0]
                    anchors.remove(old)
                    break
        new = (summit, lb, rb)
        anchors.add(new)
        for b in range(lb, rb + 1):
            owner[b] = new
    return anchors


def _components(pts, eps):
    """Labels of the connected components of the distance <= eps graph; isolated points get -1."""
    m = len(pts)
    parent = list(range(m))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    e2 = eps * eps
    P = np.asarray(pts, dtype=np.int64)
    order = np.argsort(P[:, 0], kind='stable')
    xs = P[order, 0]
    linked = np.zeros(m, dtype=bool)
    for a in range(m):
        ia = order[a]
        b = a + 1
        while b < m and xs[b] - xs[a] <= eps:
            ib = order[b]
            dx = P[ib, 0] - P[ia, 0]
            dy = P[ib, 1] - P[ia, 1]
            if dx * dx + dy * dy <= e2:
                linked[ia] = linked[ib] = True
                ra, rb = find(ia), find(ib)
                if ra != rb:
                    parent[rb] = ra
            b += 1
    return np.array([find(i) if linked[i] else -1 for i in range(m)], dtype=np.int64)


def _dist(a, b):
    return np.sqrt(float((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2))


def _cluster_core(sort_list, r, visited, final_list):
    """callers.py:636-678.  `sort_list` = [(value, (i, j))] sorted by value, descending."""
    if len(sort_list) < 2:
        return
    pts = [p[1] for p in sort_list]
    labels = _components(pts, r)
    pos = np.asarray(pts, dtype=np.int64)
    pool = set()
    for i, (_, px) in enumerate(sort_list):
        if px in pool or labels[i] == -1:
            continue
        sub = [tuple(int(v) for v in q) for q in pos[labels == labels[i]]]
        cen, rad = px, r
        local = [px]
        last_out = -1
        while len(sub):
            out = []
            for q in sub:
                if q in pool:
                    continue
                if _dist(q, cen) <= rad:
                    local.append(q)
                else:
                    out.append(q)
            if len(out) == last_out:
                break
            last_out = len(out)
            cen = tuple(int(v) for v in np.r_[local].mean(axis=0).round().astype(int))
            rad = int(np.round(max(_dist(cen, q) for q in local))) + r
            sub = out
        pool.update(local)
        final_list.append((px, cen, rad))
    visited.update(pool)


def local_clustering(Donuts, LL, res, onlysummit=False, min_count=3, r=20000, sumq=1):
    """callers.py:680-728.  Returns [(pixel, centroid, radius)] in bins."""
    final_list = []
    keys = list(Donuts)
    if not keys:
        return final_list
    x = np.r_[[k[0] for k in keys]]
    y = np.r_[[k[1] for k in keys]]
    x_anchors = find_anchors(x, min_count=min_count, min_dis=r, res=res)
    y_anchors = find_anchors(y, min_count=min_count, min_dis=r, res=res)
    r = max(r // res, 1)
    visited = set()
    lookup = set(keys)
    # The pixels inside every (x anchor, y anchor) rectangle.  The anchors of an axis do not overlap (find_anchors merges those
    # that do), so a pixel lies in at most one rectangle: the pixels are dealt to their rectangles once, and the rectangles are
    # visited in the reference's order (callers.py:700-706 walks all pairs and tests every cell of each).
    xown, yown, disjoint = {}, {}, True
    for own, anchors in ((xown, x_anchors), (yown, y_anchors)):
        for a in anchors:
            for b in range(a[1], a[2] + 1):
                disjoint = disjoint and b not in own
                own[b] = a
    disjoint = disjoint and not _WALK_ALL_CELLS
    if disjoint:
        boxes = {}
        for k in keys:
            xa, ya = xown.get(k[0]), yown.get(k[1])
            if xa is not None and ya is not None:
                boxes.setdefault((xa, ya), []).append((Donuts[k][0], k))
    for xa in x_anchors:
        for ya in y_anchors:
            if disjoint:
                inside = boxes.get((xa, ya))
                if inside is None or len(inside) < 2:       # (_cluster_core leaves fewer than two pixels alone)
                    continue
            else:
                inside = [(Donuts[(i, j)][0], (i, j)) for i in range(xa[1], xa[2] + 1) for j in range(ya[1], ya[2] + 1)
                          if (i, j) in lookup]
            inside.sort(reverse=True)
            _cluster_core(inside, r, visited, final_list)
    rest = [(Donuts[k][0], k) for k in keys if k not in visited]
    rest.sort(reverse=True)
    _cluster_core(rest, r, visited, final_list)

    x_summits = set(a[0] for a in x_anchors)
    y_summits = set(a[0] for a in y_anchors)
    for k in keys:
        if k in visited:
            continue
        if LL is not None:
            qpass = Donuts[k][-1] + LL[k][-1] <= sumq
        else:
            qpass = Donuts[k][-1] <= sumq / 2
        if qpass and ((not onlysummit) or (k[0] in x_summits) or (k[1] in y_summits)):
            final_list.append((k, k, 0))
    return final_list
