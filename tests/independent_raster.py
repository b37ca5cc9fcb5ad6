"""An independently written brute-force rasterizer for the integer raster rules (test infrastructure).

Written from the rule texts, not from oracle/oracle.cpp or the HIP kernels:

* WebGPU "Rasterization" / Vulkan "Basic Polygon Rasterization" (SURVEY.md 8c): pixel (i, j) is sampled at its centre
  (i + 1/2, j + 1/2) in a y-down framebuffer, vertex positions are first snapped to a fixed-point grid
  (subPixelPrecisionBits = 8 here), a polygon is front-facing with `frontFace = ccw` when its vertices run
  counter-clockwise on screen, back faces and zero-area polygons are culled (pipeline3d.rs:454-460).
* The top-left fill rule as Direct3D 11 words it (the rule WebGPU and Vulkan implementations follow): a sample exactly on
  an edge belongs to the triangle only if that edge is a TOP edge (exactly horizontal, with the rest of the triangle
  below it) or a LEFT edge (not horizontal, with the rest of the triangle to its right).
* Depth32Float with compare Less against the 1.0 clear, nearest fragment wins, ties to the first drawn
  (pipeline3d.rs:461-467).

The geometric formulation below (which side of an edge the OPPOSITE VERTEX lies on) shares no expression with the
edge-function form the oracle and the kernels use (rule 6); only the f32 arithmetic that DESIGN.md fixes as
the contract — the transform order of rule 1, the snap of rule 4 and the depth plane of rule 7 — is taken from there,
because "depth bits equal" is only meaningful against one stated sequence of f32 operations.

Everything here is numpy f32 (one IEEE operation per numpy call: no contraction) and Python / int64 integers.
Only triangles that need no clipping are supported (all vertices inside the view volume's depth range and the
guard band); the caller checks that with `needs_clipper`.
"""
import numpy as np

F = np.float32
SUB = 256            # 8 sub-pixel bits
GUARD = F(16.0)      # guard band half-extent in NDC units (rule 3)


def _row(m, r, x, y, z, w=None):
    """((m[r,0]*x + m[r,1]*y) + m[r,2]*z) + m[r,3]*w in f32, the order DESIGN.md rule 1 states (w None: + m[r,3])."""
    acc = (m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z
    return acc + (m[r, 3] if w is None else m[r, 3] * w)


def clip_positions(projview, model, positions):
    """vs_main (render_mesh.wgsl:42-56): world = model * (p, 1); clip = (proj * view) * world.  Row-major 4x4 f32 in."""
    pv, m = np.asarray(projview, F), np.asarray(model, F)
    p = np.asarray(positions, F)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    wx, wy, wz, ww = (_row(m, r, x, y, z) for r in range(4))
    return np.stack([_row(pv, r, wx, wy, wz, ww) for r in range(4)], axis=1).astype(F)


def needs_clipper(clip):
    """True per vertex where rule 3 would send the triangle to the clipper (or drop it)."""
    x, y, z, w = (clip[:, k] for k in range(4))
    g = GUARD * w
    return (~np.isfinite(clip).all(axis=1)) | (z < 0) | (w - z < 0) | (x + g < 0) | (g - x < 0) | (y + g < 0) | (g - y < 0) | (w <= 0)


def snap(clip, width, height):
    """Viewport transform and fixed-point snap (rule 4): integers in 1/256 px, depth z/w in f32."""
    x, y, z, w = (clip[:, k] for k in range(4))
    rw = F(1.0) / w
    sx = (x * rw + F(1.0)) * F(0.5 * width)
    sy = (F(1.0) - y * rw) * F(0.5 * height)
    X = np.rint(sx * F(SUB)).astype(np.int64)        # round half to even, as IEEE rint
    Y = np.rint(sy * F(SUB)).astype(np.int64)
    return X, Y, (z * rw).astype(F)


def _side(px, py, qx, qy, rx, ry):
    """> 0, 0, < 0: on which side of the directed line P -> Q the point R lies (exact integers)."""
    return (qx - px) * (ry - py) - (qy - py) * (rx - px)


def _edge_is_top_or_left(px, py, qx, qy, ox, oy):
    """The Direct3D wording, literally, for the edge PQ of a triangle whose third vertex is O (y grows downwards)."""
    if py == qy:                                 # exactly horizontal: a top edge iff the triangle hangs below it
        return oy > py
    # not horizontal: a left edge iff the triangle lies to its right, i.e. the third vertex is to the right of the
    # edge's line at the third vertex's height.  x on the line at y = oy is px + (qx - px) * (oy - py) / (qy - py).
    num = (qx - px) * (oy - py)                  # compare ox - px with num / den without dividing
    den = qy - py
    lhs = (ox - px) * den
    return lhs > num if den > 0 else lhs < num


def rasterize(X, Y, Z, triangles, width, height, first_id=0):
    """Brute force over every pixel centre of every triangle's bounding box.

    X, Y: snapped vertex coordinates (int64, 1/256 px); Z: z/w (f32); triangles: (n, 3) vertex indices.
    Returns (ids uint32 (h, w) with 0xFFFFFFFF where nothing is drawn, depth f32 (h, w) with 1.0 there, n_edge_ties).
    """
    ids = np.full((height, width), 0xFFFFFFFF, np.uint32)
    depth = np.full((height, width), 1.0, F)
    ties = 0
    for t, (a, b, c) in enumerate(np.asarray(triangles).reshape(-1, 3)):
        ax, ay, bx, by, cx, cy = int(X[a]), int(Y[a]), int(X[b]), int(Y[b]), int(X[c]), int(Y[c])
        # signed double area in the y-down framebuffer: positive = clockwise on screen.  ccw front faces, back faces culled.
        area2 = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay)
        if area2 >= 0:
            continue
        i0 = max((min(ax, bx, cx) - SUB // 2 + SUB - 1) // SUB, 0)
        i1 = min((max(ax, bx, cx) - SUB // 2) // SUB, width - 1)
        j0 = max((min(ay, by, cy) - SUB // 2 + SUB - 1) // SUB, 0)
        j1 = min((max(ay, by, cy) - SUB // 2) // SUB, height - 1)
        if i0 > i1 or j0 > j1:
            continue
        pxs = np.arange(i0, i1 + 1, dtype=np.int64) * SUB + SUB // 2
        pys = np.arange(j0, j1 + 1, dtype=np.int64) * SUB + SUB // 2
        gx, gy = np.meshgrid(pxs, pys)
        inside = np.ones(gx.shape, bool)
        # an edge, its opposite vertex: the sample must lie strictly on the opposite vertex's side, or exactly on the
        # edge when the edge is a top or a left one
        for (px, py, qx, qy, ox, oy) in ((ax, ay, bx, by, cx, cy), (bx, by, cx, cy, ax, ay), (cx, cy, ax, ay, bx, by)):
            want = _side(px, py, qx, qy, ox, oy)           # nonzero: the triangle has area
            s = (qx - px) * (gy - py) - (qy - py) * (gx - px)
            same = (s > 0) if want > 0 else (s < 0)
            on = s == 0
            if _edge_is_top_or_left(px, py, qx, qy, ox, oy):
                same = same | on
            ties += int((on & inside).sum())
            inside &= same
        if not inside.any():
            continue
        # depth plane (rule 7): barycentric weights of v1 and v2 from integer sub-triangle areas, f32 from there on.
        # weight of v1 = area(p, v0 -> ... ) / area: the sub-triangle opposite v1 is (v0, v2, p); of v2: (v1, v0, p);
        # both taken positive for a front-facing triangle.
        w1 = (cx - ax) * (gy - ay) - (cy - ay) * (gx - ax)      # (v2 - v0) x (p - v0)
        w2 = (ax - bx) * (gy - by) - (ay - by) * (gx - bx)      # (v0 - v1) x (p - v1)
        inv = F(1.0) / F(-area2)
        l1 = w1.astype(F) * inv
        l2 = w2.astype(F) * inv
        z0, z1, z2 = F(Z[a]), F(Z[b]), F(Z[c])
        z = (l1 * (z1 - z0) + l2 * (z2 - z0)) + z0
        z = np.minimum(np.maximum(z, F(0.0)), F(1.0)).astype(F)
        ok = inside & (z < F(1.0))
        jj, ii = np.nonzero(ok)
        jj += j0
        ii += i0
        zz = z[ok]
        old = depth[jj, ii]
        # Less: strictly nearer wins; equal depth keeps the triangle drawn first (triangles come in draw order)
        win = zz < old
        depth[jj[win], ii[win]] = zz[win]
        ids[jj[win], ii[win]] = first_id + t
    return ids, depth, ties
