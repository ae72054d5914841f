"""numpy restatement of the op-level hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  Arithmetic is done in the dtype of the inputs (fp32 ops
are individually rounded - numpy never contracts to FMA), so the integer
by-products (floor indices, mip levels) are what an IEEE, non-contracted fp32
evaluation of the reference formulas yields.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which
oracle/make_golden.py produced by importing the reference's own CPU bodies.
"""
import math

import numpy as np

# ----------------------------------------------------------------------------
# a1  upfirdn2d   (models/stylegan2/op/upfirdn2d.py:147-200, upfirdn2d_kernel.cu:108-207)
# ----------------------------------------------------------------------------

def upfirdn2d_out_size(in_h, in_w, kh, kw, up, down, pad):
    """Output size, upfirdn2d.py:105-106.  pad = (pad_x0, pad_x1, pad_y0, pad_y1)."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    return out_h, out_w


def upfirdn2d(x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """x (N,C,H,W), k (kh,kw).  Zero-stuff by `up`, pad/crop, correlate with the
    flipped kernel, decimate by `down` - upfirdn2d.py:161-200 (upfirdn2d_native).
    Tap accumulation order is ky-major, kx-minor as upfirdn2d_kernel.cu:191-196."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    n, c, h, w = x.shape
    kh, kw = k.shape
    u = np.zeros((n, c, h * up_y, w * up_x), dtype=x.dtype)
    u[:, :, ::up_y, ::up_x] = x
    u = np.pad(u, ((0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))))
    u = u[:, :, max(-py0, 0):u.shape[2] - max(-py1, 0), max(-px0, 0):u.shape[3] - max(-px1, 0)]
    hp, wp = u.shape[2], u.shape[3]
    fh, fw = hp - kh + 1, wp - kw + 1
    if fh <= 0 or fw <= 0:
        oh, ow = upfirdn2d_out_size(h, w, kh, kw, up, down, pad)
        return np.zeros((n, c, max(oh, 0), max(ow, 0)), dtype=x.dtype)
    kf = k[::-1, ::-1].astype(x.dtype)
    full = np.zeros((n, c, fh, fw), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            full += u[:, :, ky:ky + fh, kx:kx + fw] * kf[ky, kx]
    return np.ascontiguousarray(full[:, :, ::down_y, ::down_x])


def upfirdn2d_grad_pad(in_h, in_w, kh, kw, up, down, pad):
    """g_pad of the backward op, upfirdn2d.py:113-118."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    out_h, out_w = upfirdn2d_out_size(in_h, in_w, kh, kw, up, down, pad)
    g_pad_x0 = kw - px0 - 1
    g_pad_y0 = kh - py0 - 1
    g_pad_x1 = in_w * up_x - out_w * down_x + px0 - up_x + 1
    g_pad_y1 = in_h * up_y - out_h * down_y + py0 - up_y + 1
    return (g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)


def upfirdn2d_backward(grad_out, k, up, down, pad, in_size):
    """d/dx of upfirdn2d: the same op with up<->down, flipped taps, g_pad
    (UpFirDn2dBackward.forward, upfirdn2d.py:21-62)."""
    kh, kw = k.shape
    g_pad = upfirdn2d_grad_pad(in_size[2], in_size[3], kh, kw, up, down, pad)
    return upfirdn2d(grad_out, np.ascontiguousarray(k[::-1, ::-1]), up=down, down=up, pad=g_pad)


# ----------------------------------------------------------------------------
# a2  fused bias + leaky relu   (fused_bias_act_kernel.cu:18-49, fused_act.py:20-97)
# ----------------------------------------------------------------------------

def fused_bias_act(x, b=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """Generic kernel body, fused_bias_act_kernel.cu:26-47.  b indexes dim 1."""
    dt = x.dtype
    v = x
    if b is not None and b.size:
        shape = [1, -1] + [1] * (x.ndim - 2)
        v = x + b.reshape(shape).astype(dt)
    alpha = dt.type(alpha)
    scale = dt.type(scale)
    if act == 1:
        y = v if grad in (0, 1) else np.zeros_like(v)
    elif act == 3:
        if grad == 0:
            y = np.where(v > 0, v, v * alpha)
        elif grad == 1:
            y = np.where(ref > 0, v, v * alpha)
        else:
            y = np.zeros_like(v)
    else:
        raise ValueError(act)
    return (y * scale).astype(dt)


def fused_leaky_relu(x, b, negative_slope=0.2, scale=2 ** 0.5):
    """FusedLeakyReLUFunction.forward, fused_act.py:54-61."""
    return fused_bias_act(x, b, None, 3, 0, negative_slope, scale)


def fused_leaky_relu_backward(grad_out, out, negative_slope=0.2, scale=2 ** 0.5):
    """FusedLeakyReLUFunctionBackward.forward, fused_act.py:22-40: grad wrt input uses the
    saved OUTPUT as sign reference; grad wrt bias sums over every dim but 1."""
    gx = fused_bias_act(grad_out, None, out, 3, 1, negative_slope, scale)
    dims = (0,) + tuple(range(2, gx.ndim))
    gb = gx.astype(np.float64).sum(axis=dims).astype(gx.dtype)
    return gx, gb


# ----------------------------------------------------------------------------
# a11  splat2d   (utils/splat2d_cuda/src/splat_gpu_impl.cu:41-96, splat_gpu.c:12-42)
# ----------------------------------------------------------------------------

def splat2d(inp, coords, values, sigma, soft_normalize=False):
    """inp (N,C,H,W) fp32, coords (N,P,2) (x,y), values (N,P,C), sigma (N,).
    Sums are accumulated in float64 (the reference uses fp32 atomics in arbitrary
    order, so any fp32 order is legal; float64 is the order-free centre)."""
    n, c, h, w = inp.shape
    p = coords.shape[1]
    f32 = np.float32
    out = inp.astype(np.float64).copy()
    alpha_sum = np.zeros((n, h, w), dtype=np.float64)
    for i in range(n):
        stdev = f32(sigma[i])
        length = f32(2) * stdev
        # normalizer = -pow(2*stdev*stdev, -1)   splat_gpu_impl.cu:73
        normalizer = -(f32(1) / (f32(2) * stdev * stdev))
        for j in range(p):
            xc = f32(coords[i, j, 0])
            yc = f32(coords[i, j, 1])
            if not (xc >= 0 and xc < w and yc >= 0 and yc < h):
                continue
            t = int(max(f32(0), np.floor(yc - length)))
            b = int(min(f32(h - 1), np.ceil(yc + length)))
            l = int(max(f32(0), np.floor(xc - length)))
            r = int(min(f32(w - 1), np.ceil(xc + length)))
            if b < t or r < l:
                continue
            lh = np.arange(t, b + 1, dtype=f32)[:, None]
            lw = np.arange(l, r + 1, dtype=f32)[None, :]
            d2 = (lw - xc) * (lw - xc) + (lh - yc) * (lh - yc)
            a = np.exp((normalizer * d2).astype(f32)).astype(f32)
            alpha_sum[i, t:b + 1, l:r + 1] += a
            for ch in range(c):
                out[i, ch, t:b + 1, l:r + 1] += (a * f32(values[i, j, ch])).astype(f32)
    alpha32 = alpha_sum.astype(f32)
    if soft_normalize:
        alpha32 = np.maximum(alpha32, f32(1.0))
    return (out.astype(f32) / (alpha32[:, None] + f32(1e-8))).astype(f32)


# ----------------------------------------------------------------------------
# ATen grid_sample / affine_grid / interpolate (un-vendored dependency: torch>=1.10.1,
# pinned here against torch 2.10 CPU).  Call sites: antialiased_sampling.py:16,177;
# warping_heads.py:135,176,250; ATen/native/GridSampler.h:27-160.
# ----------------------------------------------------------------------------

def _unnormalize(g, size):
    dt = g.dtype.type
    return ((g + dt(1)) * dt(size) - dt(1)) / dt(2)        # GridSampler.h:27-35, align_corners=False


def _reflect(v, twice_low, twice_high):
    dt = v.dtype.type
    if twice_low == twice_high:
        return np.zeros_like(v)
    mn = dt(twice_low) / dt(2)
    span = dt(twice_high - twice_low) / dt(2)
    a = np.abs(v - mn)
    extra = np.fmod(a, span)
    flips = np.floor(a / span)
    even = np.fmod(flips, dt(2)) == 0
    return np.where(even, extra + mn, span - extra + mn).astype(v.dtype)   # GridSampler.h:89-105


def grid_source_coords(grid, h, w, padding_mode):
    """Un-normalised, padded source coordinates (ix, iy) of F.grid_sample(align_corners=False)."""
    ix = _unnormalize(grid[..., 0], w)
    iy = _unnormalize(grid[..., 1], h)
    dt = ix.dtype.type
    if padding_mode == 'border':
        ix = np.clip(ix, dt(0), dt(w - 1))
        iy = np.clip(iy, dt(0), dt(h - 1))
    elif padding_mode == 'reflection':
        ix = np.clip(_reflect(ix, -1, 2 * w - 1), dt(0), dt(w - 1))
        iy = np.clip(_reflect(iy, -1, 2 * h - 1), dt(0), dt(h - 1))
    elif padding_mode != 'zeros':
        raise ValueError(padding_mode)
    return ix, iy


def grid_sample(x, grid, padding_mode='border', return_indices=False):
    """Bilinear F.grid_sample(x, grid, padding_mode, align_corners=False).
    x (N,C,H,W), grid (N,Ho,Wo,2).  Taps outside the image contribute zero."""
    n, c, h, w = x.shape
    ix, iy = grid_source_coords(grid, h, w, padding_mode)
    dt = x.dtype.type
    x0f = np.floor(ix)
    y0f = np.floor(iy)
    x0 = x0f.astype(np.int64)
    y0 = y0f.astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    wx1 = ix - x0f
    wx0 = dt(1) - wx1
    wy1 = iy - y0f
    wy0 = dt(1) - wy1
    out = np.zeros((n, c) + ix.shape[1:], dtype=x.dtype)
    bidx = np.arange(n)[:, None, None]
    for (yy, xx, ww) in ((y0, x0, wy0 * wx0), (y0, x1, wy0 * wx1), (y1, x0, wy1 * wx0), (y1, x1, wy1 * wx1)):
        valid = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        xc = np.clip(xx, 0, w - 1)
        yc = np.clip(yy, 0, h - 1)
        vals = x[bidx, :, yc, xc]                       # (N,Ho,Wo,C)
        vals = np.where(valid[..., None], vals, dt(0)) * ww[..., None]
        out += np.moveaxis(vals, -1, 1)
    if return_indices:
        return out, x0.astype(np.int32), y0.astype(np.int32)
    return out


def linspace_aten(start, end, steps, dtype=np.float32):
    """torch.linspace's symmetric evaluation (ATen RangeFactories): step=(end-start)/(steps-1);
    first half start+i*step, second half end-(steps-1-i)*step."""
    dt = np.dtype(dtype).type
    if steps == 1:
        return np.array([start], dtype=dtype)
    step = (dt(end) - dt(start)) / dt(steps - 1)
    i = np.arange(steps)
    lo = dt(start) + step * i.astype(dtype)
    hi = dt(end) - step * (steps - 1 - i).astype(dtype)
    return np.where(i < steps // 2, lo, hi).astype(dtype)


def affine_grid(theta, out_h, out_w):
    """F.affine_grid(theta (N,2,3), (N,C,H,W), align_corners=False) -> (N,H,W,2).
    Base coordinate j -> linspace(-1,1,W)[j]*(W-1)/W  (ATen AffineGridGenerator.cpp)."""
    dt = theta.dtype
    xs = linspace_aten(-1, 1, out_w, dt) * dt.type(out_w - 1) / dt.type(out_w)
    ys = linspace_aten(-1, 1, out_h, dt) * dt.type(out_h - 1) / dt.type(out_h)
    base = np.stack([np.broadcast_to(xs[None, :], (out_h, out_w)),
                     np.broadcast_to(ys[:, None], (out_h, out_w)),
                     np.ones((out_h, out_w), dtype=dt)], axis=-1)      # (H,W,3)
    return np.einsum('hwk,nck->nhwc', base, theta).astype(dt)


def interpolate_bilinear(x, scale):
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False) with the
    scale factor used directly (recompute_scale_factor unset): src = (dst+0.5)/scale-0.5 clamped at 0
    (ATen UpSample.h area_pixel_compute_source_index)."""
    n, c, h, w = x.shape
    oh, ow = int(math.floor(h * scale)), int(math.floor(w * scale))
    dt = x.dtype.type

    def taps(o, size):
        src = (np.arange(o, dtype=x.dtype) + dt(0.5)) * dt(1.0 / scale) - dt(0.5)
        src = np.maximum(src, dt(0))
        i0 = np.minimum(src.astype(np.int64), size - 1)
        i1 = i0 + (i0 < size - 1)
        l1 = src - i0.astype(x.dtype)
        return i0, i1, dt(1) - l1, l1

    y0, y1, wy0, wy1 = taps(oh, h)
    x0, x1, wx0, wx1 = taps(ow, w)
    rows0 = x[:, :, y0, :]
    rows1 = x[:, :, y1, :]

    def hblend(r):
        return r[:, :, :, x0] * wx0 + r[:, :, :, x1] * wx1

    return (wy0[:, None] * hblend(rows0) + wy1[:, None] * hblend(rows1)).astype(x.dtype)


# ----------------------------------------------------------------------------
# a6  MipmapWarp   (models/spatial_transformers/antialiased_sampling.py:19-238)
# ----------------------------------------------------------------------------

BLUR_1331 = (np.outer([1., 3., 3., 1.], [1., 3., 3., 1.]) / 64.0).astype(np.float32)   # :101-107


def mip_downsample_2x(x):
    """ReflectionPad2d(1) + depthwise 4x4 [1,3,3,1]^2/64 stride 2 (antialiased_sampling.py:111-117)."""
    p = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='reflect')
    h, w = x.shape[2] // 2, x.shape[3] // 2
    out = np.zeros(x.shape[:2] + (h, w), dtype=x.dtype)
    for ky in range(4):
        for kx in range(4):
            out += p[:, :, ky:ky + 2 * h:2, kx:kx + 2 * w:2] * x.dtype.type(BLUR_1331[ky, kx])
    return out


def mip_pyramid(x, num_levels):
    """[x, down(x), down(down(x)), ...] - the un-upsampled Gaussian pyramid."""
    levels = [x]
    for _ in range(1, num_levels):
        levels.append(mip_downsample_2x(levels[-1]))
    return levels


def mip_stack(x, num_levels):
    """Gaussian stack (N,C,D,H,W): level i = bilinear-upsample(down^i(x), 2^i)
    (antialiased_sampling.py:119-149), including the non-power-of-two reflect pad/crop."""
    size = x.shape[-1]
    log_size = np.log2(size)
    pad_needed = not float(log_size).is_integer()
    if pad_needed:
        target = int(2 ** np.ceil(log_size))
        total = target - size
        lp = int(total // 2)
        rp = int(total - lp)
        x = np.pad(x, ((0, 0), (0, 0), (lp, rp), (lp, rp)), mode='reflect')
    pyr = mip_pyramid(x, num_levels)
    levels = [pyr[0]] + [interpolate_bilinear(pyr[i], 2.0 ** i) for i in range(1, num_levels)]
    stack = np.stack(levels, axis=2)
    if pad_needed:
        stack = stack[:, :, :, lp:-rp, lp:-rp]
    return stack


def mip_levels(grid, h, w, max_num_levels):
    """Per-pixel mip level (antialiased_sampling.py:62-97,181-210): absolute coords with the
    (size-1) convention, replicate-padded 4-neighbour max distance (squared distance clamped
    at 1 BEFORE the sqrt), log2, clamp to [0, max_num_levels-1]."""
    dt = grid.dtype.type
    xc = dt(w - 1.0) * (grid[..., 0] + dt(1)) / dt(2)
    yc = dt(h - 1.0) * (grid[..., 1] + dt(1)) / dt(2)
    c = np.stack([xc, yc], axis=-1)                                   # (N,Ho,Wo,2)
    cp = np.pad(c, ((0, 0), (1, 1), (1, 1), (0, 0)), mode='edge')
    neigh = (cp[:, 1:-1, :-2], cp[:, 1:-1, 2:], cp[:, :-2, 1:-1], cp[:, 2:, 1:-1])   # l, r, u, d
    dmax = None
    for o in neigh:
        d = o - c
        sq = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
        dist = np.sqrt(np.maximum(sq, dt(1)))
        dmax = dist if dmax is None else np.maximum(dmax, dist)
    lv = np.log2(dmax).astype(grid.dtype)
    return np.clip(lv, dt(0), dt(max_num_levels - 1.0)), dmax


def mipmap_warp(x, grid, max_num_levels=3.5, min_level=0.0, padding_mode='border', return_aux=False):
    """MipmapWarp.forward (antialiased_sampling.py:35-60)."""
    n, c, h, w = x.shape
    dt = x.dtype.type
    levels, dmax = mip_levels(grid, h, w, max_num_levels)
    levels = np.maximum(levels, dt(min_level))
    num_levels = int(np.ceil(levels.max())) + 1
    stack = mip_stack(x, num_levels)                                  # (N,C,D,H,W)
    d = stack.shape[2]
    warped = grid_sample(stack.reshape(n, c * d, h, w), grid, padding_mode)
    warped = warped.reshape((n, c, d) + warped.shape[2:])
    l0 = np.floor(levels).astype(np.int64)
    l1 = np.ceil(levels).astype(np.int64)
    wgt = np.fmod(levels, dt(1))
    idx0 = np.broadcast_to(l0[:, None, None], (n, c, 1) + l0.shape[1:])
    idx1 = np.broadcast_to(l1[:, None, None], (n, c, 1) + l1.shape[1:])
    o0 = np.take_along_axis(warped, idx0, axis=2)[:, :, 0]
    o1 = np.take_along_axis(warped, idx1, axis=2)[:, :, 0]
    out = (o0 + wgt[:, None] * (o1 - o0)).astype(x.dtype)
    if return_aux:
        ix, iy = grid_source_coords(grid, h, w, padding_mode)
        aux = dict(levels=levels, level_floor=l0.astype(np.int32), level_ceil=l1.astype(np.int32),
                   num_levels=num_levels, ix_nw=np.floor(ix).astype(np.int32),
                   iy_nw=np.floor(iy).astype(np.int32))
        return out, aux
    return out


def mip_level_ints(dmax, max_num_levels=3.5, min_level=0.0):
    """floor/ceil of the level computed from the exponent of the distance (no log2 rounding
    involved): what the HIP kernel does.  Valid for dmax >= 1."""
    top = max_num_levels - 1.0
    m, e = np.frexp(dmax.astype(np.float64))                          # dmax = m*2^e, m in [0.5,1)
    fl = (e - 1).astype(np.int64)
    exact = (m == 0.5)
    ce = np.where(exact, fl, fl + 1)
    over = np.log2(dmax.astype(np.float64)) >= top
    fl = np.where(over, int(math.floor(top)), fl)
    ce = np.where(over, int(math.ceil(top)), ce)
    fl = np.maximum(fl, int(math.floor(min_level)))
    ce = np.maximum(ce, int(math.ceil(min_level)))
    return fl.astype(np.int32), ce.astype(np.int32)


# ----------------------------------------------------------------------------
# a7/a8  similarity matrix, flow composition   (warping_heads.py:36-56,173-203,268-277)
# ----------------------------------------------------------------------------

def make_affine_matrix(params):
    """params (N,4) raw (rot, scale, tx, ty) -> (N,2,3)  (warping_heads.py:36-50)."""
    dt = params.dtype.type
    rot = np.tanh(params[:, 0]) * dt(math.pi)
    s = np.exp(params[:, 1])
    cs, sn = np.cos(rot), np.sin(rot)
    m = np.stack([s * cs, -s * sn, params[:, 2], s * sn, s * cs, params[:, 3]], axis=1)
    return m.reshape(-1, 2, 3).astype(params.dtype)


def compose_affine(base, m):
    """base (N,2,3) @ [m;0 0 1]  (warping_heads.py:52-56,120-123)."""
    n = m.shape[0]
    m3 = np.concatenate([m, np.broadcast_to(np.array([[[0, 0, 1]]], dtype=m.dtype), (n, 1, 3))], axis=1)
    return np.matmul(base, m3).astype(m.dtype)


def convex_upsample_flow(flow, mask, ds=8):
    """RAFT convex upsampling, warping_heads.py:180-193.  flow (N,h,w,2), mask (N,9*ds*ds,h,w)
    -> (N, ds*h, ds*w, 2):  up[n,ds*y+i,ds*x+j,c] = sum_k softmax_k(mask[n,k*ds*ds+i*ds+j,y,x]) *
    ds*flow0[n,y+ky-1,x+kx-1,c], k=3*ky+kx, flow zero-padded."""
    n, h, w, _ = flow.shape
    dt = flow.dtype.type
    m = mask.reshape(n, 9, ds, ds, h, w)
    m = m - m.max(axis=1, keepdims=True)
    e = np.exp(m)
    sm = e / e.sum(axis=1, keepdims=True)
    f = np.pad(flow * dt(ds), ((0, 0), (1, 1), (1, 1), (0, 0)))
    up = np.zeros((n, ds, ds, h, w, 2), dtype=flow.dtype)
    for ky in range(3):
        for kx in range(3):
            k = 3 * ky + kx
            nb = f[:, ky:ky + h, kx:kx + w, :]                        # (N,h,w,2)
            up += sm[:, k][..., None] * nb[:, None, None]
    # (N,i,j,y,x,c) -> (N,y,i,x,j,c)
    return np.ascontiguousarray(up.transpose(0, 3, 1, 4, 2, 5)).reshape(n, ds * h, ds * w, 2)


def identity_flow(size, dtype=np.float32):
    """FlowHead.initialize_flow, warping_heads.py:173-178: affine_grid(eye) -> (1,size,size,2)."""
    eye = np.array([[[1, 0, 0], [0, 1, 0]]], dtype=dtype)
    return affine_grid(eye, size, size)


def apply_affine(matrix, grid):
    """[grid,1] @ matrix^T per sample, warping_heads.py:268-277."""
    gx, gy = grid[..., 0], grid[..., 1]
    m = matrix[:, None, None]
    ox = gx * m[..., 0, 0] + gy * m[..., 0, 1] + m[..., 0, 2]
    oy = gx * m[..., 1, 0] + gy * m[..., 1, 1] + m[..., 1, 2]
    return np.stack([ox, oy], axis=-1).astype(grid.dtype)


def flow_compose(low_flow, mask, base_warp=None, ds=8):
    """FlowHead.forward's grid build (warping_heads.py:239-243): returns (flow, delta_flow)."""
    delta = convex_upsample_flow(low_flow, mask, ds)
    flow = identity_flow(delta.shape[1], delta.dtype) + delta
    if base_warp is not None:
        flow = apply_affine(base_warp, flow)
    return flow.astype(delta.dtype), delta


# ----------------------------------------------------------------------------
# a9  BilinearDownsample   (antialiased_sampling.py:241-256)
# ----------------------------------------------------------------------------

def bilinear_downsample(x, stride):
    """Reflect-pad stride//2, separable tent [1,3,..,3,1]/sum, horizontal (stride along W) then
    vertical (stride along H) depthwise conv."""
    k = np.arange(1, 2 * stride + 1, 2)
    k = np.concatenate((k, k[::-1])).astype(np.float64)
    k = (k / k.sum()).astype(np.float32).astype(x.dtype)
    r = stride // 2
    p = np.pad(x, ((0, 0), (0, 0), (r, r), (r, r)), mode='reflect')
    hp, wp = p.shape[2], p.shape[3]
    ow = (wp - 2 * stride) // stride + 1
    oh = (hp - 2 * stride) // stride + 1
    tmp = np.zeros(p.shape[:3] + (ow,), dtype=x.dtype)
    for t in range(2 * stride):
        tmp += p[:, :, :, t:t + stride * ow:stride] * k[t]
    out = np.zeros(p.shape[:2] + (oh, ow), dtype=x.dtype)
    for t in range(2 * stride):
        out += tmp[:, :, t:t + stride * oh:stride, :] * k[t]
    return out


# ----------------------------------------------------------------------------
# a10  flow regularisers   (models/losses/loss.py:4-18)
# ----------------------------------------------------------------------------

def total_variation_loss(delta_flow):
    def huber_mean(a):
        a = np.abs(a)
        return np.where(a <= 1.0, 0.5 * a * a, a - 0.5).astype(np.float64).mean()
    dy = huber_mean(delta_flow[:, :-1] - delta_flow[:, 1:])
    dx = huber_mean(delta_flow[:, :, :-1] - delta_flow[:, :, 1:])
    return np.float32(dx + dy)


def flow_identity_loss(delta_flow):
    return np.float32((delta_flow.astype(np.float64) ** 2).mean())


# ----------------------------------------------------------------------------
# a3/a4  convolutions and the modulated convolution   (conv2d_gradfix.py:22-75, networks.py:233-282)
# ----------------------------------------------------------------------------

def conv2d(x, w, bias=None, stride=1, padding=0, groups=1):
    """F.conv2d semantics (cross-correlation), float64 accumulation."""
    n, cin, h, wd = x.shape
    cout, cin_g, kh, kw = w.shape
    assert cin == cin_g * groups and cout % groups == 0
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding))).astype(np.float64)
    oh = (h + 2 * padding - kh) // stride + 1
    ow = (wd + 2 * padding - kw) // stride + 1
    out = np.zeros((n, cout, oh, ow), dtype=np.float64)
    cog = cout // groups
    for g in range(groups):
        xs = xp[:, g * cin_g:(g + 1) * cin_g]
        ws = w[g * cog:(g + 1) * cog].astype(np.float64)
        for ky in range(kh):
            for kx in range(kw):
                patch = xs[:, :, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride]
                out[:, g * cog:(g + 1) * cog] += np.einsum('nchw,oc->nohw', patch, ws[:, :, ky, kx])
    if bias is not None:
        out += bias.reshape(1, -1, 1, 1)
    return out.astype(x.dtype)


def conv_transpose2d(x, w, bias=None, stride=1, padding=0, groups=1):
    """F.conv_transpose2d semantics; w (Cin, Cout/groups, kh, kw)."""
    n, cin, h, wd = x.shape
    cin_w, cog, kh, kw = w.shape
    assert cin == cin_w
    cig = cin // groups
    oh = (h - 1) * stride - 2 * padding + kh
    ow = (wd - 1) * stride - 2 * padding + kw
    full = np.zeros((n, cog * groups, (h - 1) * stride + kh, (wd - 1) * stride + kw), dtype=np.float64)
    for g in range(groups):
        xs = x[:, g * cig:(g + 1) * cig].astype(np.float64)
        ws = w[g * cig:(g + 1) * cig].astype(np.float64)
        for ky in range(kh):
            for kx in range(kw):
                contrib = np.einsum('nchw,co->nohw', xs, ws[:, :, ky, kx])
                full[:, g * cog:(g + 1) * cog, ky:ky + stride * h:stride, kx:kx + stride * wd:stride] += contrib
    out = full[:, :, padding:padding + oh, padding:padding + ow]
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1)
    return out.astype(x.dtype)


def modulated_conv2d(x, weight, style, demodulate=True, upsample=False, blur_kernel=None, blur_pad=None):
    """ModulatedConv2d.forward (networks.py:233-282) in its per-sample-weight form.
    x (N,Cin,H,W); weight (Cout,Cin,k,k) (the module's weight[0]); style (N,Cin) AFTER the
    modulation EqualLinear.  The fp16 `normalize` branch (:237-242) is off on the hot path."""
    n, cin, h, wd = x.shape
    cout, _, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    w = (scale * weight.astype(np.float64))[None] * style.astype(np.float64)[:, None, :, None, None]
    if demodulate:
        demod = 1.0 / np.sqrt((w ** 2).sum(axis=(2, 3, 4)) + 1e-8)
        w = w * demod[:, :, None, None, None]
    w = w.astype(x.dtype)
    if upsample:
        wt = w.transpose(0, 2, 1, 3, 4).reshape(n * cin, cout, k, k)
        out = conv_transpose2d(x.reshape(1, n * cin, h, wd), wt, stride=2, padding=0, groups=n)
        out = out.reshape(n, cout, out.shape[2], out.shape[3])
        out = upfirdn2d(out, blur_kernel, pad=(blur_pad[0], blur_pad[1], blur_pad[0], blur_pad[1]))
    else:
        out = conv2d(x.reshape(1, n * cin, h, wd), w.reshape(n * cout, cin, k, k), padding=k // 2, groups=n)
        out = out.reshape(n, cout, h, wd)
    return out


def max_pool2x2(x, return_code=False):
    """2x2 / stride-2 max pooling with ATen's max_pool2d rule (torchvision's VGG16 `features`, reference
    models/losses/lpips_backbones.py:101-121): the window is scanned row-major and a later element replaces the
    running maximum only if it is strictly greater or NaN; odd trailing rows / columns are dropped (floor mode).
    -> out (N, C, H//2, W//2) [, code 0..3 = winner's position dy * 2 + dx]."""
    x = np.asarray(x)
    n, c, h, w = x.shape
    oh, ow = h // 2, w // 2
    win = np.stack([x[:, :, 0:2 * oh:2, 0:2 * ow:2], x[:, :, 0:2 * oh:2, 1:2 * ow:2],
                    x[:, :, 1:2 * oh:2, 0:2 * ow:2], x[:, :, 1:2 * oh:2, 1:2 * ow:2]], 0)
    out = win[0].copy()
    code = np.zeros(out.shape, np.uint8)
    for k in range(1, 4):
        take = (win[k] > out) | np.isnan(win[k])
        out = np.where(take, win[k], out)
        code = np.where(take, np.uint8(k), code)
    return (out, code) if return_code else out


def max_pool2x2_backward(grad_out, code, in_hw):
    """Gradient of max_pool2x2: each output's gradient goes to the winner's position, everything else is zero."""
    g = np.asarray(grad_out)
    n, c, oh, ow = g.shape
    dx = np.zeros((n, c) + tuple(in_hw), g.dtype)
    for k in range(4):
        dy_, dx_ = divmod(k, 2)
        dx[:, :, dy_:2 * oh:2, dx_:2 * ow:2] = np.where(code == k, g, 0)
    return dx
