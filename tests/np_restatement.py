"""Independent numpy/float32 restatement of the cull predicates.

Written directly from the GLSL (shaders/entity_cull.comp, meshlet_cull.comp,
depth_reduce.comp in the reference repo), vectorised over all invocations, to
cross-check the C oracle (oracle/orbit_oracle.c): two restatements by different
means must agree bit-for-bit on visibility.  Every arithmetic step is a single
float32 numpy ufunc (no einsum/dot), so there is no contraction or
reassociation.  Scans are restated with plain python loops (small cases only).
"""
import numpy as np

from orbit_amd import layouts as L

F = np.float32
S = 32

_C = np.array([float.fromhex(h) for h in (
    "0x1.715476p+0", "-0x1.71547p-1", "0x1.ec708p-2", "-0x1.715a68p-2", "0x1.2782e6p-2", "-0x1.eac694p-3",
    "0x1.a265fcp-3", "-0x1.865ffcp-3", "0x1.80ab18p-3", "-0x1.cebep-4")], dtype=np.float32)


def log2c(x):
    """The canonical software log2, vectorised."""
    x = np.asarray(x, dtype=np.float32).copy()
    out = np.empty_like(x)
    b = x.view(np.uint32)
    absb = b & np.uint32(0x7FFFFFFF)
    nan = absb > np.uint32(0x7F800000)
    zero = absb == 0
    neg = (b & np.uint32(0x80000000)) != 0
    inf = b == np.uint32(0x7F800000)
    sub = (b < np.uint32(0x00800000)) & ~zero & ~neg
    xs = np.where(sub, x * F(16777216.0), x).astype(np.float32)
    bs = xs.view(np.uint32)
    e = (bs >> np.uint32(23)).astype(np.int32) - 127 + np.where(sub, -24, 0)
    m = ((bs & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)
    big = m > F(1.41421354)
    m = np.where(big, m * F(0.5), m).astype(np.float32)
    e = e + big
    f = (m - F(1.0)).astype(np.float32)
    p = np.full_like(f, _C[9])
    for i in range(8, -1, -1):
        p = (p * f).astype(np.float32)
        p = (p + _C[i]).astype(np.float32)
    r = (p * f).astype(np.float32)
    out[:] = (r + e.astype(np.float32)).astype(np.float32)
    out[zero] = -np.inf
    out[neg & ~zero] = np.nan
    out[inf] = np.inf
    out[nan] = x[nan]
    return out


def gmax(x, y):
    return np.where(x < y, y, x).astype(np.float32)


def gmin(x, y):
    return np.where(y < x, y, x).astype(np.float32)


def dot3(ax, ay, az, bx, by, bz):
    return ((ax * bx + ay * by).astype(np.float32) + (az * bz).astype(np.float32)).astype(np.float32)


def mat_vec(m, v0, v1, v2, v3):
    """GLSL mat4 * vec4 for batched column-major m[..., 16] -> 4 arrays."""
    out = []
    for r in range(4):
        t = (m[..., 0 + r] * v0 + m[..., 4 + r] * v1).astype(np.float32)
        t = (t + (m[..., 8 + r] * v2).astype(np.float32)).astype(np.float32)
        t = (t + (m[..., 12 + r] * v3).astype(np.float32)).astype(np.float32)
        out.append(t)
    return out


def mat_mul(a, b):
    """a[16] (single) x b[n,16] -> [n,16], column-major."""
    a = np.broadcast_to(np.asarray(a, dtype=np.float32), b.shape)
    out = np.empty_like(b)
    for c in range(4):
        col = mat_vec(a, b[..., 4 * c + 0], b[..., 4 * c + 1], b[..., 4 * c + 2], b[..., 4 * c + 3])
        for r in range(4):
            out[..., 4 * c + r] = col[r]
    return out


def fma32(a, b, c):
    """fma(a, b, c) rounded once to binary32 (what fmaf / v_fma_f32 compute): the product is exact in binary64, the
    sum is rounded to 53 bits and then to 24 — the second rounding is repaired where the 53-bit sum sits exactly half
    way between two binary32 values (TwoSum gives the sign of what the first rounding dropped)."""
    a, b, c = (np.asarray(t, dtype=np.float32).astype(np.float64) for t in np.broadcast_arrays(a, b, c))
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    r = s.astype(np.float32)
    half = (np.ascontiguousarray(s).view(np.uint64) & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
    fix = half & (err != 0) & np.isfinite(s) & (np.abs(s) >= 2.0 ** -126)
    if np.any(fix):
        r64 = r.astype(np.float64)
        down = np.where(r64 > s, np.nextafter(r, np.float32(-np.inf)), r)
        up = np.where(r64 > s, r, np.nextafter(r, np.float32(np.inf)))
        r = np.where(fix, np.where(err > 0, up, down), r)
    return r.astype(np.float32)


RCP127 = np.frombuffer(np.array([0x3C010204], np.uint32).tobytes(), np.float32)[0]  # the compiled shaders' 1/127


def transform_sphere(m, sph):
    one = np.ones(len(sph), dtype=np.float32)
    p = mat_vec(m, sph[:, 0], sph[:, 1], sph[:, 2], one)
    with np.errstate(all="ignore"):
        x, y, z = (p[0] / p[3]).astype(F), (p[1] / p[3]).astype(F), (p[2] / p[3]).astype(F)
    dx = dot3(m[:, 0], m[:, 1], m[:, 2], m[:, 0], m[:, 1], m[:, 2])
    dy = dot3(m[:, 4], m[:, 5], m[:, 6], m[:, 4], m[:, 5], m[:, 6])
    dz = dot3(m[:, 8], m[:, 9], m[:, 10], m[:, 8], m[:, 9], m[:, 10])
    scale = np.sqrt(gmax(dx, gmax(dy, dz)), dtype=np.float32)
    return x, y, z, (sph[:, 3] * scale).astype(np.float32), scale


def plane_test(ci, x, y, z, r):
    vis = np.ones(len(x), dtype=bool)
    for i in range(int(ci["cull_plane_count"])):
        pl = ci["cull_planes"][i]
        d = (dot3(pl[0], pl[1], pl[2], x, y, z) + pl[3]).astype(np.float32)
        vis &= d > -r
    return vis


def f2i_clamp(f, hi):
    f = np.asarray(f, dtype=np.float32)
    out = np.zeros(f.shape, dtype=np.int64)
    ok = f >= 0
    big = ok & (f > np.float32(hi))
    mid = ok & ~big
    out[big] = hi
    out[mid] = f[mid].astype(np.int64)
    return out


def pyramid_levels(w0, h0):
    mips = max(1, int(np.floor(np.log2(max(w0, h0)))) + 1)
    offs, ws, hs, off = [], [], [], 0
    for k in range(mips):
        w, h = max(w0 >> k, 1), max(h0 >> k, 1)
        offs.append(off)
        ws.append(w)
        hs.append(h)
        off += w * h
    return mips, np.array(offs), np.array(ws), np.array(hs), off


def footprint_min(img, w, h, u, v):
    """img flat; w, h, u, v arrays (per sample)."""
    wf, hf = w.astype(np.float32), h.astype(np.float32)
    x = ((u * wf).astype(np.float32) - F(0.5)).astype(np.float32)
    y = ((v * hf).astype(np.float32) - F(0.5)).astype(np.float32)
    fx, fy = np.floor(x), np.floor(y)

    def cl(f, hi):
        f = np.asarray(f, dtype=np.float32)
        out = np.zeros(f.shape, dtype=np.int64)
        ok = f >= 0
        big = ok & (f > hi.astype(np.float32))
        mid = ok & ~big
        out[big] = hi[big]
        out[mid] = f[mid].astype(np.int64)
        return out
    x0, x1 = cl(fx, w - 1), cl((fx + F(1)).astype(F), w - 1)
    y0, y1 = cl(fy, h - 1), cl((fy + F(1)).astype(F), h - 1)
    return x0, x1, y0, y1


def hiz_sample(pyr, w0, h0, u, v, lod):
    mips, offs, ws, hs, _ = pyramid_levels(w0, h0)
    with np.errstate(all="ignore"):
        lf = (np.ceil((lod + F(0.5)).astype(F)) - F(1.0)).astype(np.float32)
    level = f2i_clamp(lf, mips - 1)
    w, h, off = ws[level], hs[level], offs[level]
    x0, x1, y0, y1 = footprint_min(pyr, w, h, u, v)
    a, b = pyr[off + y0 * w + x0], pyr[off + y0 * w + x1]
    c, d = pyr[off + y1 * w + x0], pyr[off + y1 * w + x1]
    return gmin(gmin(a, b), gmin(c, d))


def occlusion_test(ci, x, y, z, r, pyr, pw, ph, radius, scale):
    """Returns (visible, z') — z' is the possibly flipped z (persists).  Operation by operation as the reference's
    compiled shaders hold it (oracle/orbit_oracle.c occlusion_test): radius = model-space radius, r = radius * scale."""
    n = len(x)
    with np.errstate(all="ignore"):
        if int(ci["projection_type"]) == 0:
            z = (-z).astype(np.float32)
            zn = F(ci["z_near"])
            cullable = z >= fma32(radius, scale, zn)
            p00, p11 = F(ci["p00_or_width_recipx2"]), F(ci["p11_or_height_recipx2"])

            def bounds(c0, c1):
                # cx = -C.xz ; vx = (sqrt(dot(cx,cx) - r*r), r)
                vx = np.sqrt(fma32(-r, r, (c0 * c0 + c1 * c1).astype(F)), dtype=F)
                vy = r
                mn_x = ((vx * c0).astype(F) + ((-vy) * c1).astype(F)).astype(F)
                mn_y = ((vy * c0).astype(F) + (vx * c1).astype(F)).astype(F)
                mx_x = ((vx * c0).astype(F) + (vy * c1).astype(F)).astype(F)
                mx_y = (((-vy) * c0).astype(F) + (vx * c1).astype(F)).astype(F)
                return mn_x, mn_y, mx_x, mx_y
            minx_x, minx_y, maxx_x, maxx_y = bounds((-x).astype(F), (-z).astype(F))
            miny_x, miny_y, maxy_x, maxy_y = bounds((-y).astype(F), (-z).astype(F))
            a0 = ((minx_x / minx_y).astype(F) * p00).astype(F)
            a1 = ((miny_x / miny_y).astype(F) * p11).astype(F)
            a2 = ((maxx_x / maxx_y).astype(F) * p00).astype(F)
            a3 = ((maxy_x / maxy_y).astype(F) * p11).astype(F)
            u0 = fma32(a0, F(0.5), F(0.5))
            v0 = fma32(a3, F(-0.5), F(0.5))
            u1 = fma32(a2, F(0.5), F(0.5))
            v1 = fma32(a1, F(-0.5), F(0.5))
            closest = (zn / fma32(-radius, scale, z)).astype(F)
        else:
            sr = F(ci["p00_or_width_recipx2"])
            cx, cy = (x * sr).astype(F), (y * sr).astype(F)
            bs = (sr * r).astype(F)
            b0, b1 = fma32(bs, F(-1.0), cx), fma32(bs, F(-1.0), cy)
            b2, b3 = fma32(bs, F(1.0), cx), fma32(bs, F(1.0), cy)
            cl = lambda t: gmin(gmax(t, F(-1.0)), F(1.0))
            u0 = fma32(cl(b0), F(0.5), F(0.5))
            v0 = fma32(cl(b1), F(-0.5), F(0.5))
            u1 = fma32(cl(b2), F(0.5), F(0.5))
            v1 = fma32(cl(b3), F(-0.5), F(0.5))
            cullable = np.ones(n, dtype=bool)
            rr = F(1.0) / (F(ci["z_far"]) - F(ci["z_near"]))
            closest = (rr * (fma32(radius, scale, z) + F(ci["z_far"])).astype(F)).astype(F)
        width = ((u1 - u0).astype(F) * F(pw)).astype(F)
        height = ((v1 - v0).astype(F) * F(ph)).astype(F)
        u = ((u0 + u1).astype(F) * F(0.5)).astype(F)
        v = ((v0 + v1).astype(F) * F(0.5)).astype(F)
        sampled = hiz_sample(pyr, pw, ph, u, v, log2c(gmax(width, height)))
        vis = np.where(cullable, closest >= sampled, True)
    return vis, z


def f2u_sat(f):
    f = np.asarray(f, dtype=np.float32)
    out = np.zeros(f.shape, dtype=np.uint64)
    pos = f > 0
    big = pos & (f >= F(4294967296.0))
    mid = pos & ~big
    out[big] = 0xFFFFFFFF
    out[mid] = f[mid].astype(np.uint64)
    return out


def entity_cull(ci, scene_draws, count, entity_draw_count, mesh_infos, entities, vis_words, pyr=None, pyr_size=(0, 0)):
    """Returns (visible[g], should_draw[g], records list, new entity words or None)."""
    end = min(count, (entity_draw_count + 255) // 256 * 256)
    draws = scene_draws[:end]
    g = np.arange(end)
    mi = mesh_infos[draws["mesh_index"]]
    en = entities[draws["entity_index"]]
    op = int(ci["occlusion_pass"])
    meshlet_occ = int(ci["meshlet_visibility_buffer"]) != L.NONE
    vib = np.ones(end, dtype=bool)
    if op in (1, 2):
        vib = ((vis_words[g // 32] >> (g % 32).astype(np.uint32)) & 1).astype(bool)
    visible = vib.copy() if op == 1 else np.ones(end, dtype=bool)
    mv = mat_mul(ci["view_matrix"], en["model_matrix"])
    x, y, z, r, scale = transform_sphere(mv, mi["bounding_sphere"])
    visible &= np.where(visible, plane_test(ci, x, y, z, r), False)
    if op == 2:
        ov, zf = occlusion_test(ci, x, y, z, r, pyr, *pyr_size, mi["bounding_sphere"][:, 3].astype(F), scale)
        z = np.where(visible, zf, z).astype(np.float32) if int(ci["projection_type"]) == 0 else z
        visible = np.where(visible, ov, False)
    should = visible.copy()
    if op == 2:
        should = visible & (~vib | meshlet_occ)
    t = ci["lod_target_pos_view_space"]
    ex, ey, ez = (t[0] - x).astype(F), (t[1] - y).astype(F), (t[2] - z).astype(F)
    with np.errstate(all="ignore"):
        dist = (np.sqrt(dot3(ex, ey, ez, ex, ey, ez), dtype=F) - r).astype(F)
        lf = (log2c((gmax(dist, F(0.0)) / F(ci["lod_base"])).astype(F)) / log2c(np.array([ci["lod_step"]], F))).astype(F)
        lod = f2u_sat(gmax((lf + F(1.0)).astype(F), F(0.0)))
    lod = np.minimum(np.maximum(lod, int(ci["min_mesh_lod"])), int(ci["max_mesh_lod"]))
    lod = np.minimum(lod, (mi["lod_count"].astype(np.uint64) - 1) & 0xFFFFFFFF)
    lod = np.minimum(lod, 7).astype(np.int64)
    records = []
    for i in np.nonzero(should)[0]:
        off, cnt = mi["mesh_lods"][i, lod[i]]
        vo = int(draws["visibility_offset"][i])
        for j in range((int(cnt) + S - 1) // S):
            c = min(int(cnt) - S * j, S)
            records.append((int(draws["entity_index"][i]), int(off) + S * j, c, vo))
            vo += c // S
    new_words = None
    if op == 2:
        new_words = vis_words.copy()
        for w in range((end + 31) // 32):
            bits = visible[32 * w:32 * w + 32]
            new_words[w] = sum(int(b) << k for k, b in enumerate(bits))
    return visible, should, np.array(records, dtype=np.uint32).reshape(-1, 4).view(L.MESHLET_DISPATCH).reshape(-1), new_words


def meshlet_cull(ci, records, meshlets, entities, materials, mvis, pyr=None, pyr_size=(0, 0)):
    """Returns (commands, new meshlet visibility words or None)."""
    op = int(ci["occlusion_pass"])
    meshlet_occ = int(ci["meshlet_visibility_buffer"]) != L.NONE
    nrec = len(records)
    lane = np.tile(np.arange(S), nrec)
    rid = np.repeat(np.arange(nrec), S)
    active = lane < records["meshlet_count"][rid]
    rid, lane = rid[active], lane[active]
    rec = records[rid]
    idx = rec["meshlet_offset"].astype(np.int64) + lane
    m = meshlets[idx]
    mv = mat_mul(ci["view_matrix"], entities["model_matrix"][rec["entity_index"]])
    x, y, z, r, scale = transform_sphere(mv, m["bounding_sphere"])
    ax = (m["cone_axis"].astype(np.int32).astype(np.float32) * RCP127).astype(F)
    axis = mat_vec(mv, ax[:, 0], ax[:, 1], ax[:, 2], np.zeros(len(ax), F))
    cutoff = (m["cone_cutoff"].astype(np.int32).astype(np.float32) * RCP127).astype(F)
    alpha = materials["alpha_mode"][m["material_index"]]
    rw = op in (1, 2) and meshlet_occ
    vib = np.ones(len(idx), dtype=bool)
    if rw:
        vib = ((mvis[rec["visibility_offset"] + lane // 32] >> (lane % 32).astype(np.uint32)) & 1).astype(bool)
    visible = vib.copy() if op == 1 else np.ones(len(idx), dtype=bool)
    visible &= plane_test(ci, x, y, z, r)
    with np.errstate(all="ignore"):
        if int(ci["projection_type"]) == 1:
            camx, camy, camz = (x - F(0)).astype(F), (y - F(0)).astype(F), (z - F(-1.0)).astype(F)
        else:
            camx = camy = camz = np.zeros(len(x), F)
        dx, dy, dz = (x - camx).astype(F), (y - camy).astype(F), (z - camz).astype(F)
        lhs = dot3(dx, dy, dz, axis[0], axis[1], axis[2])
        rhs = fma32(cutoff, np.sqrt(dot3(dx, dy, dz, dx, dy, dz), dtype=F), r)
        visible &= ~(lhs >= rhs)
    if meshlet_occ and op == 2:
        ov, _ = occlusion_test(ci, x, y, z, r, pyr, *pyr_size, m["bounding_sphere"][:, 3].astype(F), scale)
        visible = np.where(visible, ov, False)
    shl = lambda a: np.where(a < 32, np.uint64(1) << a.astype(np.uint64), 0).astype(np.uint64)
    should = visible & ((shl(alpha) & np.uint64(int(ci["alpha_mode_flag"]))) != 0)
    if op == 2 and meshlet_occ:
        skip = (shl(alpha) & np.uint64(int(ci["noskip_alphamode"]))) != 0
        should = np.where(~skip, visible & ~vib, should)
    cmds = np.zeros(int(should.sum()), dtype=L.MESHLET_DRAW_COMMAND)
    ms = m[should]
    cmds["cmd_index_count"] = ms["triangle_count"].astype(np.uint32) * 3
    cmds["cmd_instance_count"] = 1
    cmds["cmd_first_index"] = (ms["data_offset"] + ms["vertex_count"].astype(np.uint32)) * np.uint32(4)
    cmds["cmd_vertex_offset"] = ms["data_offset"].view(np.int32)
    cmds["cmd_first_instance"] = rec["entity_index"][should]
    cmds["meshlet_vertex_offset"] = ms["vertex_offset"]
    cmds["meshlet_index"] = idx[should].astype(np.uint32)
    new = None
    if op == 2 and meshlet_occ:
        new = mvis.copy()
        words = {}
        for ri, ln, vb in zip(rid, lane, visible):
            w = int(records["visibility_offset"][ri]) + ln // 32
            words[w] = words.get(w, 0) | (int(vb) << (ln % 32))
        for w, val in words.items():
            new[w] = val
    return cmds, new


def depth_reduce(depth, sw, sh):
    from math import floor, log2

    def npot(v):
        p = 1
        while p < v:
            p <<= 1
        return p
    w0, h0 = max(npot(sw) // 2, 1), max(npot(sh) // 2, 1)
    mips, offs, ws, hs, total = pyramid_levels(w0, h0)
    pyr = np.zeros(total, dtype=np.float32)
    src, srcw, srch = np.ascontiguousarray(depth, np.float32).reshape(-1), sw, sh
    for k in range(mips):
        dw, dh = int(ws[k]), int(hs[k])
        ys, xs = np.mgrid[0:dh, 0:dw]
        u = ((xs.astype(F) + F(0.5)) / F(dw)).astype(F).reshape(-1)
        v = ((ys.astype(F) + F(0.5)) / F(dh)).astype(F).reshape(-1)
        W = np.full(u.shape, srcw)
        H = np.full(u.shape, srch)
        x0, x1, y0, y1 = footprint_min(src, W, H, u, v)
        a, b = src[y0 * srcw + x0], src[y0 * srcw + x1]
        c, d = src[y1 * srcw + x0], src[y1 * srcw + x1]
        lvl = gmin(gmin(a, b), gmin(c, d))
        pyr[offs[k]:offs[k] + dw * dh] = lvl
        src, srcw, srch = lvl, dw, dh
    return pyr, (w0, h0, mips)
