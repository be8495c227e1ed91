"""Seeded input generators shared by the CPU and GPU tests."""
import numpy as np

from deftet_amd import grids


def jittered(res, n_query, batch, jitter=0.1):
    tet, pts, tets, nv = grids.make_case(res, n_query, batch, jitter)
    return tet, pts


def adversarial(seed=0, n_extra_query=600):
    """One shape mixing a regular jittered grid with every special class the binned
    algorithm treats separately: flat / tiny / inverted / duplicated / huge / non-finite
    tets, and queries on vertices, on faces, far away, huge, inf and NaN."""
    rng = np.random.default_rng(seed)
    tet, pts = jittered(6, 300, 1, 0.15)
    tet = tet[0].copy()
    T0 = tet.shape[0]
    extra = []
    # inverted copies of some tets (swap two vertices) — all sign_v false, still "regular"
    inv = tet[rng.integers(0, T0, 20)].copy()
    inv[:, [0, 1]] = inv[:, [1, 0]]
    extra.append(inv)
    # exact duplicates (lowest index must win)
    extra.append(tet[rng.integers(0, T0, 20)].copy())
    # flat tets (four coplanar points) and needle tets (two coincident vertices)
    flat = rng.uniform(-0.5, 0.5, (20, 4, 3)).astype(np.float32)
    flat[:, 3] = (flat[:, 0] + flat[:, 1] + flat[:, 2]) / 3
    extra.append(flat)
    needle = rng.uniform(-0.5, 0.5, (10, 4, 3)).astype(np.float32)
    needle[:, 3] = needle[:, 2]
    extra.append(needle)
    # fully collapsed tets (all four vertices equal): every dotp is +-0 -> accepts everything
    pt = rng.uniform(-0.5, 0.5, (3, 1, 3)).astype(np.float32)
    extra.append(np.repeat(pt, 4, axis=1))
    # slivers: tiny volume relative to extent
    sl = rng.uniform(-0.5, 0.5, (20, 4, 3)).astype(np.float32)
    sl[:, 3] = (sl[:, 0] + sl[:, 1] + sl[:, 2]) / 3 + rng.normal(0, 1e-6, (20, 3)).astype(np.float32)
    extra.append(sl)
    # tiny and huge tets
    extra.append((rng.uniform(-1, 1, (10, 4, 3)) * 1e-12).astype(np.float32))
    extra.append((rng.uniform(-1, 1, (6, 4, 3)) * 1e7).astype(np.float32))
    big = rng.uniform(-3, 3, (6, 4, 3)).astype(np.float32)        # large well-shaped tets
    extra.append(big)
    # non-finite tets
    bad = rng.uniform(-0.5, 0.5, (6, 4, 3)).astype(np.float32)
    bad[0, 0, 0] = np.nan
    bad[1, 2, 1] = np.inf
    bad[2, 3, 2] = -np.inf
    bad[3] = np.nan
    extra.append(bad)
    extra = np.concatenate(extra, 0)
    # Degenerate tets can accept whole half-spaces (all four tests "false"), so where they
    # sit decides how much of the regular machinery is visible: seed%3==0 keeps them at the
    # end (they only catch what nothing else contains), ==1 interleaves them in the second
    # half, ==2 interleaves everything.
    if seed % 3 == 0:
        all_t = np.concatenate([tet[rng.permutation(T0)], extra[rng.permutation(extra.shape[0])]], 0)
    elif seed % 3 == 1:
        half = T0 // 2
        tail = np.concatenate([tet[half:], extra], 0)
        all_t = np.concatenate([tet[:half], tail[rng.permutation(tail.shape[0])]], 0)
    else:
        all_t = np.concatenate([tet, extra], 0)
        all_t = all_t[rng.permutation(all_t.shape[0])]
    all_t = np.ascontiguousarray(all_t)

    q = [pts[0]]
    q.append(all_t[rng.integers(0, all_t.shape[0], 60), rng.integers(0, 4, 60)])          # on vertices
    tri = all_t[rng.integers(0, all_t.shape[0], 80)]
    w = rng.dirichlet([1, 1, 1], 80).astype(np.float32)
    q.append((tri[:, :3] * w[:, :, None]).sum(1))                                          # on faces
    w4 = rng.dirichlet([1, 1, 1, 1], 80).astype(np.float32)
    q.append((tri * w4[:, :, None]).sum(1))                                                # inside
    q.append(rng.uniform(-5, 5, (n_extra_query // 4, 3)).astype(np.float32))               # far
    q.append((rng.uniform(-1, 1, (20, 3)) * 1e7).astype(np.float32))                       # huge
    q.append((rng.uniform(-1, 1, (20, 3)) * 1e-9).astype(np.float32))                      # tiny
    sp = rng.uniform(-0.5, 0.5, (12, 3)).astype(np.float32)
    sp[0, 0] = np.nan; sp[1] = np.nan; sp[2, 1] = np.inf; sp[3, 2] = -np.inf; sp[4] = np.inf
    sp[5] = 0.0; sp[6] = -0.0; sp[7] = 3.0e38; sp[8, 0] = 2.0e6; sp[9, 0] = -2.0e6
    q.append(sp)
    q = np.concatenate([np.nan_to_num(x, nan=np.nan) for x in q], 0).astype(np.float32)
    q = q[rng.permutation(q.shape[0])]
    return all_t[None], np.ascontiguousarray(q[None])


def scaled(scale, offset, seed=1):
    """A jittered grid pushed to an unusual scale / offset (tests the relative margins)."""
    tet, pts = jittered(6, 400, 2, 0.1)
    off = np.asarray(offset, np.float32)
    return (tet * np.float32(scale) + off).astype(np.float32), (pts * np.float32(scale) + off).astype(np.float32)
