"""Independent NumPy restatement of the shader (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

A second, separately written reading of
/root/reference/shaders/importance_driven_volume_rendering.wgsl, vectorised over pixels in
float32, used only to cross-check oracle/volym_oracle.c on small images (SURVEY.md section 4
"oracle-vs-oracle").  It takes the same uniform bytes and returns the same outputs and fetch
counters.  Parity with the reference itself stays unpinned (oracle/volym_oracle.h).
"""
import numpy as np

F = np.float32
ZERO, ONE = F(0.0), F(1.0)

# cos/sin of (s/8)*2*3.14159, s = 0..7 (wgsl:99-103), as float32; tests re-derive them
CONE_COS = np.array([float.fromhex(h) for h in (
    "0x1p+0", "0x1.6a09f6p-1", "0x1.54442ep-20", "-0x1.6a09bap-1",
    "-0x1p+0", "-0x1.6a0a32p-1", "-0x1.fe6644p-19", "0x1.6a097ep-1")], F)
CONE_SIN = np.array([float.fromhex(h) for h in (
    "0x0p+0", "0x1.6a09d8p-1", "0x1p+0", "0x1.6a0a14p-1",
    "0x1.54442ep-19", "-0x1.6a099cp-1", "-0x1p+0", "-0x1.6a0a5p-1")], F)


# ---- elementary functions: the recipe of DESIGN.md "Elementary functions", plain f32 ops ----
def wgsl_log2(x):
    x = np.asarray(x, F)
    bits = x.view(np.uint32)
    e = (bits >> np.uint32(23)).astype(np.int32) - 127
    m = ((bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(F)
    big = m > F(1.41421356)
    m = np.where(big, m * F(0.5), m)
    e = np.where(big, e + 1, e)
    s = (m - ONE) / (m + ONE)
    s2 = s * s
    p = np.full_like(s, F(0.3205989))
    for c in (F(0.412198573), F(0.577078044), F(0.961796701), F(2.88539004)):
        p = p * s2 + c
    return e.astype(F) + s * p


def wgsl_exp2(z):
    z = np.asarray(z, F)
    zero = ~(z >= F(-126.0))
    zc = np.where(zero, ZERO, np.minimum(z, F(127.0)))
    n = np.rint(zc)
    f = zc - n
    p = np.full_like(f, F(1.52527336e-5))
    for c in (F(1.54035297e-4), F(1.33335579e-3), F(9.61812865e-3), F(5.55041097e-2), F(2.40226507e-1),
              F(6.93147182e-1), ONE):
        p = p * f + c
    scale = ((n.astype(np.int32) + 127).astype(np.uint32) << np.uint32(23)).view(F)
    return np.where(zero, ZERO, p * scale)


def wgsl_pow(x, y):
    x = np.asarray(x, F)
    y = np.broadcast_to(np.asarray(y, F), x.shape)
    safe = np.where(x > ZERO, x, ONE)
    r = wgsl_exp2(y * wgsl_log2(safe))
    r = np.where(x == ZERO, ZERO, r)
    return np.where(y == ZERO, ONE, r).astype(F)


def wgsl_exp(x):
    return wgsl_exp2(np.asarray(x, F) * F(1.44269502))


# ---- samplers ---------------------------------------------------------------------------------
def _texel_nearest(u, n):
    f = np.floor(u * F(n))
    f = np.where(f >= ZERO, f, ZERO)          # also NaN -> 0
    return np.minimum(f, F(n - 1)).astype(np.int64)


def _fetch_nearest(tex, dims, px, py, pz):
    nx, ny, nz = dims
    ix, iy, iz = _texel_nearest(px, nx), _texel_nearest(py, ny), _texel_nearest(pz, nz)
    return tex[ix + nx * (iy + ny * iz)].astype(F) / F(255.0)


def _texel_linear(u, n):
    x = u * F(n) - F(0.5)
    fl = np.floor(x)
    w = x - fl
    fl = np.where(fl >= F(-2.0), fl, F(-2.0))
    fl = np.minimum(fl, F(n))
    i = fl.astype(np.int64)
    return np.clip(i, 0, n - 1), np.clip(i + 1, 0, n - 1), w


def _fetch_linear(tex, dims, px, py, pz):
    nx, ny, nz = dims
    x0, x1, fx = _texel_linear(px, nx)
    y0, y1, fy = _texel_linear(py, ny)
    z0, z1, fz = _texel_linear(pz, nz)

    def T(x, y, z):
        return tex[x + nx * (y + ny * z)].astype(F) / F(255.0)

    c00 = T(x0, y0, z0) * (ONE - fx) + T(x1, y0, z0) * fx
    c10 = T(x0, y1, z0) * (ONE - fx) + T(x1, y1, z0) * fx
    c01 = T(x0, y0, z1) * (ONE - fx) + T(x1, y0, z1) * fx
    c11 = T(x0, y1, z1) * (ONE - fx) + T(x1, y1, z1) * fx
    c0 = c00 * (ONE - fy) + c10 * fy
    c1 = c01 * (ONE - fy) + c11 * fy
    return c0 * (ONE - fz) + c1 * fz


class _Scene:
    pass


def _sample_volume(s, px, py, pz):
    s.n_vol += int(px.size)
    if s.filter == 1:
        return _fetch_linear(s.vol, s.dims, px, py, pz)
    return _fetch_nearest(s.vol, s.dims, px, py, pz)


def _sample_importance(s, px, py, pz):
    s.n_imp += int(px.size)
    return _fetch_nearest(s.imp, s.dims, px, py, pz)


def _sample_tf(s, u):
    i0, i1, w = _texel_linear(u, s.tf_n)
    a = s.lut[i0].astype(F) / F(255.0)
    b = s.lut[i1].astype(F) / F(255.0)
    return a * (ONE - w)[:, None] + b * w[:, None]


def _outside01(px, py, pz):
    return (px < ZERO) | (py < ZERO) | (pz < ZERO) | (px > ONE) | (py > ONE) | (pz > ONE)


def _length(x, y, z):
    return np.sqrt((x * x + y * y) + z * z)


def _normalize(x, y, z):
    with np.errstate(divide="ignore", invalid="ignore"):
        ln = _length(x, y, z)
        return x / ln, y / ln, z / ln


def _smoothed(s, px, py, pz, dx, dy, dz):      # wgsl:52-75
    sm = np.zeros_like(px)
    ws = np.zeros_like(px)
    for i in range(-2, 3):
        off = F(i) * F(0.005)
        sx, sy, sz = px + dx * off, py + dy * off, pz + dz * off
        ok = ~_outside01(sx, sy, sz)
        if ok.any():
            v = _sample_volume(s, sx[ok], sy[ok], sz[ok])
            w = s.gauss_w[i + 2]
            sm[ok] = sm[ok] + v * w
            ws[ok] = ws[ok] + w
    with np.errstate(divide="ignore", invalid="ignore"):
        return sm / ws


def _ahead_straight(s, px, py, pz, dx, dy, dz, t_exit):      # wgsl:141-160
    n = int(s.par.importance_check_ahead_steps)
    with np.errstate(divide="ignore", invalid="ignore"):
        step = (t_exit - _length(px, py, pz)) / F(n)
    found = np.zeros(px.shape, bool)
    x, y, z = px.copy(), py.copy(), pz.copy()
    for _ in range(n):
        go = ~found
        if not go.any():
            break
        x[go] = x[go] + dx[go] * step[go]
        y[go] = y[go] + dy[go] * step[go]
        z[go] = z[go] + dz[go] * step[go]
        imp = _sample_importance(s, x[go], y[go], z[go])
        found[np.flatnonzero(go)[imp >= F(0.5)]] = True
    return found


def _ahead_cone(s, px, py, pz, dx, dy, dz, t_exit):          # wgsl:94-139
    n = int(s.par.importance_check_ahead_steps)
    with np.errstate(divide="ignore", invalid="ignore"):
        step = (t_exit - _length(px, py, pz)) / F(n)
        # right = normalize(cross(d, (0,1,0))) = normalize((d.y*0 - d.z*1, d.z*0 - d.x*0, d.x*1 - d.y*0))
        rx, ry, rz = _normalize(dy * ZERO - dz * ONE, dz * ZERO - dx * ZERO, dx * ONE - dy * ZERO)
        ux, uy, uz = dy * rz - dz * ry, dz * rx - dx * rz, dx * ry - dy * rx
    found = np.zeros(px.shape, bool)
    for c in range(8):
        xo = CONE_COS[c] * F(0.2)
        yo = CONE_SIN[c] * F(0.2)
        with np.errstate(divide="ignore", invalid="ignore"):
            sx, sy, sz = _normalize((dx + rx * xo) + ux * yo, (dy + ry * xo) + uy * yo, (dz + rz * xo) + uz * yo)
        x, y, z = px.copy(), py.copy(), pz.copy()
        alive = ~found
        for _ in range(n):
            if not alive.any():
                break
            x[alive] = x[alive] + sx[alive] * step[alive]
            y[alive] = y[alive] + sy[alive] * step[alive]
            z[alive] = z[alive] + sz[alive] * step[alive]
            out = _outside01(x, y, z) & alive
            alive &= ~out
            if not alive.any():
                break
            imp = _sample_importance(s, x[alive], y[alive], z[alive])
            hit = np.flatnonzero(alive)[imp >= F(0.5)]
            found[hit] = True
            alive[hit] = False
    return found


def _shade(s, px, py, pz, cr, cg, cb):                        # wgsl:181-211
    o = F(0.01)
    two_o = F(2.0) * o
    gx = (_sample_volume(s, px + o, py, pz) - _sample_volume(s, px - o, py, pz)) / two_o
    gy = (_sample_volume(s, px, py + o, pz) - _sample_volume(s, px, py - o, pz)) / two_o
    gz = (_sample_volume(s, px, py, pz + o) - _sample_volume(s, px, py, pz - o)) / two_o
    nx, ny, nz = _normalize(gx, gy, gz)
    with np.errstate(invalid="ignore"):
        lit = _length(nx, ny, nz) > ZERO
    il = ONE / np.sqrt(F(3.0))                                 # normalize(1,1,1) = 1 / sqrt((1+1)+1)
    ex, ey, ez = _normalize(s.eye[0] - px, s.eye[1] - py, s.eye[2] - pz)
    hx, hy, hz = _normalize(ex + il, ey + il, ez + il)
    with np.errstate(invalid="ignore"):
        diffuse = np.fmax(ZERO, (nx * il + ny * il) + nz * il)
        spec = wgsl_pow(np.nan_to_num(np.fmax(ZERO, (hx * nx + hy * ny) + hz * nz), nan=0.0).astype(F), F(24.0))
    kd = F(0.2) + F(0.7) * diffuse
    ks = F(0.4) * spec
    r = np.where(lit, cr * kd + ks, cr)
    g = np.where(lit, cg * kd + ks, cg)
    b = np.where(lit, cb * kd + ks, cb)
    return r.astype(F), g.astype(F), b.astype(F)


def render(volume, importances, dims, lut, cam, par, W, H, filter=0):
    """Same contract as oracle.render: (rgba_f32 [H,W,4], rgba_u8 [H,W,4], counters dict)."""
    s = _Scene()
    s.vol = np.ascontiguousarray(volume, np.uint8).ravel()
    s.imp = np.ascontiguousarray(importances, np.uint8).ravel()
    s.dims = tuple(int(d) for d in dims)
    s.lut = np.ascontiguousarray(lut, np.uint8).reshape(-1, 4)
    s.tf_n = s.lut.shape[0]
    s.filter = int(filter)
    s.par = par
    s.n_vol = s.n_imp = 0
    s.eye = np.array(list(cam.camera_position), F)
    offs = np.arange(-2, 3).astype(F) * F(0.005)
    sigma = F(1.5)
    s.gauss_w = wgsl_exp(-(offs * offs) / (F(2.0) * sigma * sigma))
    ivp = np.array(cam.inverse_view_proj, F)                   # [col][row]
    thr = F(par.density_threshold)
    base = F(par.raymarching_step_size)
    min_step = base * F(0.25)

    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    gx = gx.ravel().astype(F)
    gy = gy.ravel().astype(F)
    n = gx.size
    ndx = (gx / F(W)) * F(2.0) - ONE                          # wgsl:221-229
    ndy = ONE - (gy / F(H)) * F(2.0)
    wp = [((ivp[0][r] * ndx + ivp[1][r] * ndy) + ivp[2][r] * ZERO) + ivp[3][r] * ONE for r in range(4)]
    with np.errstate(divide="ignore", invalid="ignore"):
        dx, dy, dz = _normalize(wp[0] / wp[3] - s.eye[0], wp[1] / wp[3] - s.eye[1], wp[2] / wp[3] - s.eye[2])
        t1 = [(ZERO - s.eye[i]) / d for i, d in enumerate((dx, dy, dz))]   # wgsl:162-179
        t2 = [(ONE - s.eye[i]) / d for i, d in enumerate((dx, dy, dz))]
    tmin = [np.fmin(a, b) for a, b in zip(t1, t2)]
    tmax = [np.fmax(a, b) for a, b in zip(t1, t2)]
    t_entry = np.fmax(np.fmax(np.fmax(tmin[0], tmin[1]), tmin[2]), ZERO)
    t_exit = np.fmax(np.fmin(np.fmin(tmax[0], tmax[1]), tmax[2]), ZERO)
    with np.errstate(invalid="ignore"):
        miss = t_exit <= t_entry

    acc = np.zeros((n, 3), F)
    acc_a = np.where(miss, ONE, ZERO).astype(F)                # wgsl:238-241
    t = t_entry.astype(F).copy()
    cur = np.full(n, base, F)
    active = ~miss
    counters = {"n_hit": int(active.sum()), "n_steps": 0, "n_dense": 0}

    while True:
        with np.errstate(invalid="ignore"):
            active &= (t < t_exit) & (acc_a < F(0.95))         # wgsl:250
        idx = np.flatnonzero(active)
        if idx.size == 0:
            break
        counters["n_steps"] += int(idx.size)
        ti = t[idx]
        px, py, pz = s.eye[0] + dx[idx] * ti, s.eye[1] + dy[idx] * ti, s.eye[2] + dz[idx] * ti
        if par.use_gaussian_smoothing == 1:
            rho = _smoothed(s, px, py, pz, dx[idx], dy[idx], dz[idx])
        else:
            rho = _sample_volume(s, px, py, pz)
        imp = _sample_importance(s, px, py, pz)                # wgsl:260
        with np.errstate(invalid="ignore"):
            dense = rho >= thr
        cur[idx] = np.where(dense, min_step, np.fmin(base, cur[idx] * F(1.5)))   # wgsl:263-269
        nd = idx[~dense]
        t[nd] = t[nd] + cur[nd]                                 # wgsl:271-274
        if not dense.any():
            continue
        d_i = idx[dense]
        counters["n_dense"] += int(d_i.size)
        px, py, pz, rho, imp = px[dense], py[dense], pz[dense], rho[dense], imp[dense]
        use_alpha = par.use_opacity == 1
        keep = np.ones(d_i.size, bool)
        if par.use_importance_coloring == 1:                   # wgsl:83-92, 279-281
            ca = np.stack([np.fmin(imp * F(1.5), ONE), (ONE - imp) * F(1.2), np.full_like(imp, F(0.2)), imp], 1)
            use_alpha = True
        else:
            if par.use_importance_rendering == 1:              # wgsl:283-295
                fn = _ahead_cone if par.use_cone_importance_check == 1 else _ahead_straight
                ahead = fn(s, px, py, pz, dx[d_i], dy[d_i], dz[d_i], t_exit[d_i])
                keep = ~((imp < ONE) & ahead)
                sk = d_i[~keep]
                t[sk] = t[sk] + cur[sk]
            ca = _sample_tf(s, rho)                             # wgsl:297-303
        if not keep.any():
            continue
        k_i = d_i[keep]
        r, g, b = _shade(s, px[keep], py[keep], pz[keep], ca[keep, 0], ca[keep, 1], ca[keep, 2])
        if use_alpha:                                           # wgsl:313-318
            alpha = ONE - wgsl_pow(ONE - ca[keep, 3], cur[k_i] * F(100.0))
            w = (ONE - acc_a[k_i]) * alpha
            acc[k_i, 0] = acc[k_i, 0] + r * w
            acc[k_i, 1] = acc[k_i, 1] + g * w
            acc[k_i, 2] = acc[k_i, 2] + b * w
            acc_a[k_i] = acc_a[k_i] + w
            t[k_i] = t[k_i] + cur[k_i]                          # wgsl:325
        else:                                                   # wgsl:319-323
            acc[k_i, 0], acc[k_i, 1], acc[k_i, 2] = r, g, b
            acc_a[k_i] = ONE
            active[k_i] = False

    f32 = np.concatenate([acc, acc_a[:, None]], 1).reshape(H, W, 4).astype(F)
    q = np.nan_to_num(f32, nan=0.0)
    u8 = np.where(q >= ONE, 255, np.where(q > ZERO, np.floor(q * F(255.0) + F(0.5)), 0)).astype(np.uint8)
    counters["n_vol"], counters["n_imp"] = s.n_vol, s.n_imp
    return f32, u8, counters
