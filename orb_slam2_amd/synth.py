"""Deterministic synthetic inputs for the ORB front end.

No dataset ships with the reference and there is no network, so every test and
benchmark runs on seeded synthetic data:

* ``textured_image``: 8-bit "textured" frame = 4-octave value noise + a few
  thousand random rectangles / rotated rectangles / discs (FAST corners at many
  scales, >= 3x oversupply for the quadtree) + +-4 pixel noise.
* ``stereo_pair``: right image = left shifted by a per-row-band disparity plus
  independent noise (gives ComputeStereoMatches real row-band matches).
* ``random_descriptors``: uniform 256-bit descriptors with planted
  near-duplicates so the best/second-best ratio test passes and fails
  non-trivially.

The generator is SplitMix64 on uint64 (self-contained, identical everywhere);
frame ``i`` of a sequence uses seed ``0x0B5EED00 + i``.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 0x0B5EED00
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class SplitMix64:
    """Vectorised SplitMix64 stream."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def next_u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = np.uint64(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, n: int) -> np.ndarray:
        """float64 in [0,1)."""
        return (self.next_u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def integers(self, lo: int, hi: int, n: int) -> np.ndarray:
        """int64 in [lo, hi)."""
        return lo + (self.next_u64(n) % np.uint64(hi - lo)).astype(np.int64)


def _value_noise(rng: SplitMix64, w: int, h: int, cell: int) -> np.ndarray:
    gw, gh = w // cell + 2, h // cell + 2
    g = rng.uniform(gw * gh).reshape(gh, gw)
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def textured_image(w: int, h: int, seed: int = BASE_SEED, shapes_per_mpx: float = 6500.0) -> np.ndarray:
    """One synthetic frame, uint8 [h, w], C-contiguous."""
    rng = SplitMix64(seed)
    img = np.zeros((h, w), dtype=np.float64)
    for cell, amp in ((64, 90.0), (32, 50.0), (16, 30.0), (8, 20.0)):
        img += amp * _value_noise(rng, w, h, cell)
    img += 30.0
    n_shapes = max(8, int(shapes_per_mpx * w * h / 1e6))
    kind = rng.integers(0, 3, n_shapes)
    cx = rng.uniform(n_shapes) * w
    cy = rng.uniform(n_shapes) * h
    # sizes: many small (fine-scale corners), some large (coarse-scale corners)
    sz = 3.0 + 40.0 * rng.uniform(n_shapes) ** 2.5
    asp = 0.4 + 1.2 * rng.uniform(n_shapes)
    ang = rng.uniform(n_shapes) * np.pi
    grey = 10.0 + 235.0 * rng.uniform(n_shapes)
    for i in range(n_shapes):
        hw, hh = sz[i], sz[i] * asp[i]
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(0, int(cx[i]) - r), min(w, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(h, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dx, dy = xx - cx[i], yy - cy[i]
        if kind[i] == 0:  # axis-aligned rectangle
            m = (np.abs(dx) <= hw) & (np.abs(dy) <= hh)
        elif kind[i] == 1:  # rotated rectangle
            ca, sa = np.cos(ang[i]), np.sin(ang[i])
            m = (np.abs(dx * ca + dy * sa) <= hw) & (np.abs(-dx * sa + dy * ca) <= hh)
        else:  # disc
            m = dx * dx + dy * dy <= hw * hw
        img[y0:y1, x0:x1][m] = grey[i]
    noise = rng.integers(-4, 5, w * h).reshape(h, w)
    out = np.clip(np.rint(img) + noise, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(out)


def frame_sequence(n: int, w: int, h: int, first: int = 0) -> np.ndarray:
    """``n`` frames [n, h, w] with seeds BASE_SEED + first + i."""
    return np.stack([textured_image(w, h, BASE_SEED + first + i) for i in range(n)])


def stereo_pair(w: int, h: int, seed: int = BASE_SEED):
    """(left, right): right(x, y) = left(x + d(y), y) for a per-row-band disparity d in [2, 60] + noise."""
    left = textured_image(w, h, seed)
    rng = SplitMix64(seed ^ 0x57E2E0)
    band = 24
    nb = (h + band - 1) // band
    d = rng.integers(2, 61, nb)
    right = np.empty_like(left)
    xs = np.arange(w)
    for b in range(nb):
        src = np.clip(xs + d[b], 0, w - 1)
        right[b * band:(b + 1) * band] = left[b * band:(b + 1) * band][:, src]
    noise = rng.integers(-2, 3, w * h).reshape(h, w)
    right = np.clip(right.astype(np.int64) + noise, 0, 255).astype(np.uint8)
    return left, np.ascontiguousarray(right)


def random_descriptors(n: int, seed: int) -> np.ndarray:
    """uint8 [n, 32] uniform random 256-bit descriptors."""
    rng = SplitMix64(seed)
    return rng.next_u64(n * 4).view(np.uint8).reshape(n, 32).copy()


def planted_queries(db: np.ndarray, nq: int, seed: int, planted_frac: float = 0.1) -> np.ndarray:
    """Queries: uniform random, except ``planted_frac`` of them are DB rows with k~U[0,60] bits flipped."""
    rng = SplitMix64(seed)
    q = random_descriptors(nq, seed ^ 0xABCDEF)
    n_pl = int(nq * planted_frac)
    if n_pl and len(db):
        rows = rng.integers(0, len(db), n_pl)
        kflip = rng.integers(0, 61, n_pl)
        for i in range(n_pl):
            v = db[rows[i]].copy()
            bits = rng.integers(0, 256, int(kflip[i]))
            for b in bits:
                v[b >> 3] ^= np.uint8(1 << (b & 7))
            q[i * (nq // n_pl)] = v
    return q


def random_vocabulary(k: int, L: int, seed: int, irregular: bool = False):
    """A synthetic DBoW2-style vocabulary tree as flat arrays (the ORBvoc.txt blob is not shipped with the reference).

    Nodes are numbered breadth first (children ids > parent id, as DBoW2's create()/loadFromTextFile produce them).
    A child's descriptor is its parent's with ~30 random bits flipped, so that descents are not arbitrary; a few siblings
    are exact duplicates (ties: the first child must win).  ``irregular``: 1..k children per node and leaves at several
    depths.  Returns child_off[n+1], child_idx, node_desc[n,32], word_id[n] (-1 for inner nodes), weight[n] (float64,
    ~5 % of the words stopped with weight 0).
    """
    rng = SplitMix64(seed)
    desc = [np.zeros(32, np.uint8)]
    depth = [0]
    children = [[]]
    frontier = [0]
    while frontier:
        nxt = []
        for p in frontier:
            if depth[p] >= L:
                continue
            if irregular and depth[p] >= 1 and int(rng.integers(0, 8, 1)[0]) == 0:
                continue                                   # an early leaf
            nc = k if not irregular else 1 + int(rng.integers(0, k, 1)[0])
            for c in range(nc):
                d = desc[p].copy()
                if depth[p] == 0:
                    d = random_descriptors(1, int(rng.next_u64(1)[0] & 0x7FFFFFFF))[0]
                else:
                    bits = rng.integers(0, 256, 30)
                    for b in bits:
                        d[b >> 3] ^= np.uint8(1 << (b & 7))
                if c > 0 and int(rng.integers(0, 16, 1)[0]) == 0:
                    d = desc[children[p][-1]].copy()       # duplicate of the previous sibling
                nid = len(desc)
                desc.append(d)
                depth.append(depth[p] + 1)
                children.append([])
                children[p].append(nid)
                nxt.append(nid)
        frontier = nxt
    n = len(desc)
    child_off = np.zeros(n + 1, np.int32)
    for i in range(n):
        child_off[i + 1] = child_off[i] + len(children[i])
    child_idx = np.array([c for ch in children for c in ch], np.int32)
    word_id = np.full(n, -1, np.int32)
    weight = np.zeros(n, np.float64)
    w = 0
    for i in range(n):
        if not children[i]:
            word_id[i] = w
            w += 1
            u = float(rng.next_u64(1)[0] >> 11) / float(1 << 53)
            weight[i] = 0.0 if u < 0.05 else 0.5 + 9.5 * u   # idf-like
    return child_off, child_idx, np.stack(desc), word_id, weight


def regular_vocabulary(k: int, L: int, seed: int):
    """Full k-ary tree of depth L, built level by level with numpy (1.1 M nodes for k=10, L=6 in a few seconds): same
    array layout as random_vocabulary, a child = its parent with 24 random bits flipped."""
    rng = SplitMix64(seed)
    levels = [np.zeros((1, 32), np.uint8)]
    for lv in range(L):
        par = np.repeat(levels[-1], k, axis=0)
        n = len(par)
        if lv == 0:
            par = random_descriptors(n, seed ^ 0x5EED)
        else:
            par = par.copy()
            for _ in range(24):
                bit = rng.integers(0, 256, n)
                par[np.arange(n), bit >> 3] ^= (1 << (bit & 7)).astype(np.uint8)
        levels.append(par)
    node_desc = np.concatenate(levels)
    n = len(node_desc)
    n_inner = n - len(levels[-1])
    child_off = np.zeros(n + 1, np.int32)
    child_off[1:n_inner + 1] = k * np.arange(1, n_inner + 1)
    child_off[n_inner + 1:] = k * n_inner
    child_idx = np.arange(1, n, dtype=np.int32)      # breadth-first numbering: node i's children are k*i+1 .. k*i+k
    word_id = np.full(n, -1, np.int32)
    word_id[n_inner:] = np.arange(n - n_inner)
    u = (rng.next_u64(n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    weight = np.where(np.arange(n) >= n_inner, np.where(u < 0.05, 0.0, 0.5 + 9.5 * u), 0.0)
    return child_off, child_idx, node_desc, word_id, weight
