"""The threaded BVH of the device library, built on the host and walked in numpy (no GPU needed).

Checks the invariants the device kernels rely on: every triangle sits in exactly one leaf, every visiting order walks
every node exactly once when nothing is culled, boxes contain their subtrees, and a stack-free walk with culling finds
the same closest hit as a scan over all triangles (the reference's semantics, hitable.rs:35-62 + geometry.rs:29-63,
later triangle wins at equal t) — for every one of the 8 direction orders, matching the ray's octant or not.
"""
import numpy as np
import pytest

F = np.float32


def soup(n, seed, flat=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-10, 10, (n, 1, 3))
    v = c + rng.uniform(-1.0, 1.0, (n, 3, 3))
    if flat:  # axis-aligned, zero-thickness boxes
        v[:, :, 1] = np.round(c[:, :, 1])
    return v.astype(F).reshape(n, 9)


def links(nodes):
    return nodes[..., 6].view(np.int32), nodes[..., 7].view(np.int32)


def tri_hit(v0, e1, e2, o, d, t_min, t_max):
    """geometry.rs:29-63 in float32, back faces culled."""
    P = np.cross(d, e2).astype(F)
    det = F(np.dot(P, e1))
    if det < 0 or det == 0:
        return None
    inv = F(1.0) / det
    T = (o - v0).astype(F)
    Q = np.cross(T, e1).astype(F)
    u = F(inv * F(np.dot(P, T)))
    if u < 0 or u > 1:
        return None
    v = F(inv * F(np.dot(Q, d)))
    if v < 0 or F(u + v) > 1:
        return None
    t = F(inv * F(np.dot(Q, e2)))
    if t < t_min or t > t_max:
        return None
    return t


def brute(tris, o, d, t_min, t_max):
    best_t, best_p = F(t_max), -1
    for k in range(tris.shape[0]):
        t = tri_hit(tris[k, 0:3], tris[k, 4:7], tris[k, 8:11], o, d, t_min, t_max)
        if t is None:
            continue
        p = int(tris[k, 3:4].view(np.int32)[0])
        if t < best_t or (t == best_t and p > best_p):
            best_t, best_p = t, p
    return best_t, best_p


def walk(nodes, tris, o, d, t_min, t_max):
    """The kernels' stack-free walk (one box test per node, culling against the closest hit so far)."""
    link, miss = links(nodes)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1.0) / d).astype(F)
        best_t, best_p, node, visited = F(t_max), -1, 0, 0
        while node >= 0:
            visited += 1
            assert visited <= nodes.shape[0], "a walk never sees a node twice"
            b = nodes[node]
            t0 = ((b[[0, 2, 4]] - o) * inv).astype(F)
            t1 = ((b[[1, 3, 5]] - o) * inv).astype(F)
            tn = np.fmax(np.fmax.reduce(np.fmin(t0, t1)), F(t_min))
            tf = np.fmin(np.fmin.reduce(np.fmax(t0, t1)), best_t)
            inside = tn <= F(tf * F(1.00001) + F(1e-7))
            lk = int(link[node])
            if inside and lk < 0:
                first, count = (~lk) >> 3, ((~lk) & 7) + 1
                for k in range(first, first + count):
                    t = tri_hit(tris[k, 0:3], tris[k, 4:7], tris[k, 8:11], o, d, t_min, t_max)
                    if t is not None:
                        p = int(tris[k, 3:4].view(np.int32)[0])
                        if t < best_t or (t == best_t and p > best_p):
                            best_t, best_p = t, p
            node = lk if (inside and lk >= 0) else int(miss[node])
    return best_t, best_p, visited


@pytest.mark.parametrize("n,ordered", [(1, True), (2, False), (7, True), (200, True), (200, False), (1500, True)])
def test_structure(rt, n, ordered):
    tri_v = soup(n, n)
    nodes, tris, n_top = rt.bvh_build_host(tri_v, max_top=16, ordered=ordered)
    assert nodes.shape[0] == (8 if ordered else 1) and 1 <= nodes.shape[1] <= 2 * n - 1 and 1 <= n_top <= min(16, nodes.shape[1])
    # triangles: a permutation of the input, edges precomputed
    prim = tris[:, 3].view(np.int32)
    assert sorted(prim.tolist()) == list(range(n))
    assert np.array_equal(tris[:, 0:3], tri_v[prim, 0:3])
    assert np.array_equal(tris[:, 4:7], tri_v[prim, 3:6] - tri_v[prim, 0:3])
    assert np.array_equal(tris[:, 8:11], tri_v[prim, 6:9] - tri_v[prim, 0:3])
    # the boxes are the same in every order; the array of octant k stores each axis' pair as (entry plane, exit plane)
    # for its rays: (max, min) along the axes whose direction component is negative (bit a of k)
    for k in range(1, nodes.shape[0]):
        for a in range(3):
            pair = nodes[k, :, 2 * a:2 * a + 2]
            want = nodes[0, :, 2 * a:2 * a + 2][:, ::-1] if (k >> a) & 1 else nodes[0, :, 2 * a:2 * a + 2]
            assert np.array_equal(pair, want)
    nodes = nodes.copy()
    for k in range(nodes.shape[0]):  # back to (min, max) for the checks below
        for a in range(3):
            nodes[k, :, 2 * a:2 * a + 2] = np.sort(nodes[k, :, 2 * a:2 * a + 2], axis=1)
    for k in range(nodes.shape[0]):
        link, miss = links(nodes[k])
        # following link at inner nodes and miss at leaves visits every node exactly once, every triangle exactly once
        seen_nodes, seen_tris, node = [], [], 0
        while node >= 0:
            seen_nodes.append(node)
            lk = int(link[node])
            if lk < 0:
                first, count = (~lk) >> 3, ((~lk) & 7) + 1
                assert 1 <= count <= 4
                seen_tris += list(range(first, first + count))
                # the leaf's box contains its (padded) triangles
                v = np.concatenate([tris[first:first + count, 0:3], tris[first:first + count, 0:3] + tris[first:first + count, 4:7],
                                    tris[first:first + count, 0:3] + tris[first:first + count, 8:11]])
                b = nodes[k, node]
                assert (v.min(0) >= b[[0, 2, 4]]).all() and (v.max(0) <= b[[1, 3, 5]]).all()
                node = int(miss[node])
            else:
                node = lk
            assert len(seen_nodes) <= nodes.shape[1]
        assert sorted(seen_nodes) == list(range(nodes.shape[1]))
        assert sorted(seen_tris) == list(range(n))
        # always taking miss from the root ends the walk at once; a child's box lies inside its parent's
        assert miss[0] == -1
        for node in range(nodes.shape[1]):
            lk = int(link[node])
            if lk >= 0:
                # children: the first one, then wherever a child's subtree exits to, until the exit is the parent's own
                c, n_children = lk, 0
                while c != int(miss[node]):
                    assert (nodes[k, c, [0, 2, 4]] >= nodes[k, node, [0, 2, 4]]).all() and (nodes[k, c, [1, 3, 5]] <= nodes[k, node, [1, 3, 5]]).all()
                    n_children += 1
                    c = int(miss[c])
                assert 2 <= n_children <= 8


@pytest.mark.parametrize("flat", [False, True])
def test_walk_matches_scan_in_every_order(rt, flat):
    n = 300
    tri_v = soup(n, 77, flat)
    nodes, tris, _ = rt.bvh_build_host(tri_v, max_top=8, ordered=True)
    rng = np.random.default_rng(5)
    hits = 0
    fewer = 0
    for r in range(60):
        o = rng.uniform(-14, 14, 3).astype(F)
        tgt = tri_v[rng.integers(n)].reshape(3, 3).mean(0) + rng.normal(0, 0.1, 3)
        d = (tgt - o).astype(F)
        if r % 10 == 0:
            d[r // 10 % 3] = F(0.0)  # a zero component: that axis' slab test degenerates
        want = brute(tris, o, d, F(1e-3), F(1e7))
        octant = int(d[0] < 0) | (int(d[1] < 0) << 1) | (int(d[2] < 0) << 2)
        visited = {}
        for k in (octant, octant ^ 7, (octant + 3) % 8):
            t, p, visited[k] = walk(nodes[k], tris, o, d, F(1e-3), F(1e7))
            assert (t, p) == want, (r, k)
        hits += want[1] >= 0
        fewer += visited[octant] <= visited[octant ^ 7]
    assert hits >= 10
    assert fewer > 40  # front-to-back order visits no more nodes than back-to-front for most rays


def test_widening_dissolves_nodes_that_cannot_cull(rt, monkeypatch):
    """A closed room of 6 quads: every inner box spans the room, so a widened tree has fewer nodes than the binary one."""
    q = []
    for axis in range(3):
        for side in (-1.0, 1.0):
            u, v = [a for a in range(3) if a != axis]
            c = np.zeros((4, 3))
            c[:, axis] = side
            c[:, u] = [-1, 1, 1, -1]
            c[:, v] = [-1, -1, 1, 1]
            q += [np.concatenate([c[0], c[1], c[2]]), np.concatenate([c[0], c[2], c[3]])]
    tri_v = np.array(q, F)
    monkeypatch.setenv("RT_BVH_ARITY", "2")
    binary, _, _ = rt.bvh_build_host(tri_v, 8, True)
    monkeypatch.setenv("RT_BVH_ARITY", "8")
    wide, tris, _ = rt.bvh_build_host(tri_v, 8, True)
    assert wide.shape[1] < binary.shape[1]
    o, d = np.array([0.1, -0.2, 0.3], F), np.array([0.3, 1.0, -0.2], F)
    want = brute(tris, o, d, F(1e-3), F(1e7))
    assert want[1] >= 0
    for k in range(8):
        t, p, visited = walk(wide[k], tris, o, d, F(1e-3), F(1e7))
        assert (t, p) == want and visited <= wide.shape[1]


@pytest.mark.parametrize("passes", ["0", "1", "8"])
def test_tree_rotations_keep_every_hit(rt, monkeypatch, passes):
    """The builder may swap subtrees after the top-down build (RT_BVH_ROTATE passes): every triangle stays in exactly
    one leaf, and a walk finds what a scan of all triangles finds, in every visiting order."""
    n = 500
    tri_v = soup(n, 123, False)
    monkeypatch.setenv("RT_BVH_ROTATE", passes)
    nodes, tris, _ = rt.bvh_build_host(tri_v, max_top=8, ordered=True)
    link = nodes[0, :, 6].view(np.int32)
    leaves = link[link < 0]
    covered = np.zeros(n, int)
    for code in ~leaves:
        first, count = int(code) >> 3, (int(code) & 7) + 1
        covered[first:first + count] += 1
    assert (covered == 1).all()
    rng = np.random.default_rng(9)
    for r in range(40):
        o = rng.uniform(-14, 14, 3).astype(F)
        d = (tri_v[rng.integers(n)].reshape(3, 3).mean(0) + rng.normal(0, 0.2, 3) - o).astype(F)
        want = brute(tris, o, d, F(1e-3), F(1e7))
        for k in (0, 5, 7):
            t, p, _ = walk(nodes[k], tris, o, d, F(1e-3), F(1e7))
            assert (t, p) == want, (r, k)


def test_tree_rotations_change_the_tree(rt, monkeypatch):
    tri_v = soup(2000, 5, False)
    monkeypatch.setenv("RT_BVH_ROTATE", "0")
    plain, _, _ = rt.bvh_build_host(tri_v, 8, False)
    monkeypatch.setenv("RT_BVH_ROTATE", "8")
    rotated, _, _ = rt.bvh_build_host(tri_v, 8, False)
    assert plain.shape != rotated.shape or not np.array_equal(plain, rotated)


def test_empty_and_argument_errors(rt):
    nodes, tris, n_top = rt.bvh_build_host(np.zeros((0, 9), F))
    assert nodes.shape[1] == 0 and tris.shape[0] == 0 and n_top == 0
    with pytest.raises(rt.RtError):
        rt.bvh_build_host(soup(3, 1), max_top=-1)


# ---- the 16-byte copy of the tree (one load per visited node) -------------------------------------------------------------
def walk_compact(cn, q, tris, octant, o, d, t_min, t_max, presorted):
    """walk_range's step (rt_device.h) in numpy: planes from 16-bit grid coordinates, entry / exit already sorted for the
    octant, first child = next node, w = where to go on after a miss."""
    n = cn.shape[0]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1.0) / d).astype(F)
        a = (q[3:6] * inv).astype(F)
        b = ((q[0:3] - o) * inv).astype(F)
        best_t, best_p, node, visited = F(t_max), -1, 0, 0
        while node < n:
            visited += 1
            assert visited <= n
            w = cn[node]
            e = ((w[:3] & 0xFFFF).astype(F) * a + b).astype(F)
            x = ((w[:3] >> 16).astype(F) * a + b).astype(F)
            if presorted:
                tn, tf = np.fmax(np.fmax.reduce(e), F(t_min)), np.fmin(np.fmin.reduce(x), best_t)
            else:
                tn, tf = np.fmax(np.fmax.reduce(np.fmin(e, x)), F(t_min)), np.fmin(np.fmin.reduce(np.fmax(e, x)), best_t)
            inside = tn <= F(tf * F(1.00001) + F(1e-7))
            leaf = bool(w[3] >> 31)
            if inside and leaf:
                code = int(w[3] & 0x7FFFFFFF)
                for k in range(code >> 3, (code >> 3) + (code & 7) + 1):
                    t = tri_hit(tris[k, 0:3], tris[k, 4:7], tris[k, 8:11], o, d, t_min, t_max)
                    if t is not None:
                        p = int(tris[k, 3:4].view(np.int32)[0])
                        if t < best_t or (t == best_t and p > best_p):
                            best_t, best_p = t, p
            node = node + 1 if (inside or leaf) else int(w[3])
    return best_t, best_p, visited


@pytest.mark.parametrize("n,ordered", [(1, True), (3, False), (200, True), (200, False), (1200, True)])
def test_compact_nodes_structure_and_walk(rt, n, ordered):
    tri_v = soup(n, n + 5, flat=(n == 200))
    cn, q = rt.bvh_build_host_compact(tri_v, ordered=ordered)
    nodes, tris, _ = rt.bvh_build_host(tri_v, max_top=0, ordered=ordered)
    assert cn.shape[0] == nodes.shape[0] == (8 if ordered else 1) and cn.shape[1] == nodes.shape[1]
    nn = cn.shape[1]
    root_lo = np.sort(nodes[0, 0, :6].reshape(3, 2), axis=1)[:, 0]
    assert np.allclose(q[:3], root_lo)
    for k in range(cn.shape[0]):
        c = cn[k]
        leaf = (c[:, 3] >> 31).astype(bool)
        skip = np.where(leaf, np.arange(nn) + 1, c[:, 3]).astype(np.int64)
        # depth-first order: every subtree is the index range [i, skip[i]); ranges nest, the root's is everything
        assert skip[0] == nn and (skip > np.arange(nn)).all() and (skip <= nn).all()
        for i in range(nn):
            if not leaf[i]:
                j = i + 1
                while j < skip[i]:  # the children: next node, then wherever each child's range ends
                    assert skip[j] <= skip[i]
                    lo_c, hi_c = np.minimum(c[j, :3] & 0xFFFF, c[j, :3] >> 16), np.maximum(c[j, :3] & 0xFFFF, c[j, :3] >> 16)
                    lo_p, hi_p = np.minimum(c[i, :3] & 0xFFFF, c[i, :3] >> 16), np.maximum(c[i, :3] & 0xFFFF, c[i, :3] >> 16)
                    assert (lo_c >= lo_p).all() and (hi_c <= hi_p).all()  # quantisation is monotone: children stay inside
                    j = skip[j]
                assert j == skip[i]
        # every triangle in exactly one leaf; a leaf's dequantised box holds its triangles; entry/exit sorted for the octant
        seen = []
        for i in np.nonzero(leaf)[0]:
            code = int(c[i, 3] & 0x7FFFFFFF)
            first, count = code >> 3, (code & 7) + 1
            seen += list(range(first, first + count))
            v = np.concatenate([tris[first:first + count, 0:3], tris[first:first + count, 0:3] + tris[first:first + count, 4:7],
                                tris[first:first + count, 0:3] + tris[first:first + count, 8:11]])
            e = q[:3] + (c[i, :3] & 0xFFFF).astype(F) * q[3:]
            x = q[:3] + (c[i, :3] >> 16).astype(F) * q[3:]
            assert (np.minimum(e, x) <= v.min(0)).all() and (np.maximum(e, x) >= v.max(0)).all()
            if ordered:
                for a in range(3):
                    assert (e[a] >= x[a]) if (k >> a) & 1 else (e[a] <= x[a])
        assert sorted(seen) == list(range(n))
    # the walk over the compact nodes finds the brute-force closest hit, in the ray's own octant (and, unsorted, in any)
    rng = np.random.default_rng(n)
    for _ in range(60 if n <= 200 else 25):
        o = rng.uniform(-14, 14, 3).astype(F)
        d = rng.normal(size=3).astype(F)
        if rng.random() < 0.2:
            d[rng.integers(3)] = F(0.0) * (F(-1.0) if rng.random() < 0.5 else F(1.0))  # +0 and -0 components
        want = brute(tris, o, d, F(1e-3), F(1e7))
        octant = int(np.signbit(d[0])) | (int(np.signbit(d[1])) << 1) | (int(np.signbit(d[2])) << 2) if ordered else 0
        got = walk_compact(cn[octant], q, tris, octant, o, d, F(1e-3), F(1e7), presorted=ordered)
        assert got[:2] == want, (o, d)
