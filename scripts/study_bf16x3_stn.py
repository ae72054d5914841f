"""CPU study (authoring container; needs /root/reference): where does the bf16x3 arithmetic lose the 1e-4 on the
similarity+flow STN's warped output?  The reference STN runs on CPU with F.conv2d replaced by an emulation of the split
precision kernels (operands split into bf16 limbs, limb-pair products i + j < LIMBS accumulated in float32), selectable
per convolution call; reports max |err| of the similarity parameters, the flow and the warped output against the
reference's own float32 run.  TEST INFRASTRUCTURE ONLY (imports oracle/)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.make_golden import import_reference            # noqa: E402
from oracle.det_weights import det_state_dict              # noqa: E402
from oracle import config_cases as cc                      # noqa: E402

REAL_CONV = F.conv2d


def limbs_of(t, n):
    out, r = [], t
    for _ in range(n):
        h = r.to(torch.bfloat16).to(torch.float32)
        out.append(h)
        r = r - h
    return out


def rtz16(x):
    """float32 -> binary16 with round-toward-zero and saturation (v_cvt_pkrtz_f16_f32), returned as float32."""
    x = x.clamp(-65504.0, 65504.0)
    h = x.to(torch.float16)
    over = h.float().abs() > x.abs()                 # nearest rounded away from zero: step the magnitude back
    bits = h.view(torch.int16).clone()
    bits[over] = bits[over] - 1
    return bits.view(torch.float16).float()


def f16_limbs_act(t):
    """Activation limbs of the fp16x3 mode: limb 0 round-toward-zero, limb 1 = nearest of the (exact) residual."""
    h0 = rtz16(t)
    return [h0, (t - h0).to(torch.float16).float()]


def f16_limbs_weight(t, scale=256.0):
    """Weight limbs of the fp16x3 packs: pre-scaled by 2^8, both limbs nearest with saturation."""
    out, r = [], t * scale
    for _ in range(2):
        h = r.clamp(-65504.0, 65504.0).to(torch.float16).float()
        out.append(h)
        r = r - h
    return out


def emulated_conv(x, w, bias, limbs, **kw):
    if limbs == 16:                                  # fp16x3: binary16 limbs, three products, accumulators / 2^8
        xs, ws = f16_limbs_act(x), f16_limbs_weight(w)
        acc = REAL_CONV(xs[1], ws[0], None, **kw) + REAL_CONV(xs[0], ws[1], None, **kw)
        acc = (acc + REAL_CONV(xs[0], ws[0], None, **kw)) / 256.0
        return acc if bias is None else acc + bias.view(1, -1, 1, 1)
    xs, ws = limbs_of(x, limbs), limbs_of(w, limbs)
    acc = None
    for s in range(limbs - 1, -1, -1):          # smallest terms first, as the kernels do
        for la in range(s + 1):
            y = REAL_CONV(xs[s - la], ws[la], None, **kw)
            acc = y if acc is None else acc + y
    if bias is not None:
        acc = acc + bias.view(1, -1, 1, 1)
    return acc


class Policy:
    """call counter + per-call limb choice; split_ok as in op/conv_mfma.py (cin % 32 == 0 and cout > 32)."""

    def __init__(self, fn):
        self.fn, self.calls, self.log = fn, 0, []

    def __call__(self, input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        x, w = input, weight
        idx = self.calls
        self.calls += 1
        limbs = self.fn(idx, tuple(x.shape), tuple(w.shape))
        ok = x.shape[1] % 32 == 0 and w.shape[0] > 32
        self.log.append((idx, tuple(x.shape), tuple(w.shape), limbs if ok else 0))
        if not ok or limbs == 0:
            return REAL_CONV(x, w, bias, stride, padding, dilation, groups)
        return emulated_conv(x, w, bias, limbs, stride=stride, padding=padding, dilation=dilation, groups=groups)


def run(policy_fn, ci=0, n=16, dtype=torch.float32):
    from models.spatial_transformers.spatial_transformer import get_stn
    from models.spatial_transformers.antialiased_sampling import BilinearDownsample
    stn = get_stn(['similarity', 'flow'], flow_size=128, supersize=256, channel_multiplier=0.5, num_heads=1)
    torch.nn.Module.load_state_dict(stn, det_state_dict(stn, cc.STN_RULES), strict=False)
    stn = stn.to(dtype)
    for m in stn.modules():
        if isinstance(m.__dict__.get('identity_flow'), torch.Tensor):
            m.identity_flow = m.identity_flow.to(dtype)
    mode, full = [('reflection', False), ('border', True)][ci]
    x = cc.smooth_images(f'c2stn.x{ci}', n, 256).to(dtype)
    small = BilinearDownsample(2, 3).to(dtype)(x)
    params = {}
    stn.stns[0].warp_head.linear.register_forward_hook(lambda m, i, o: params.__setitem__('sim', o.detach().clone()))
    pol = Policy(policy_fn) if policy_fn is not None else None
    if pol is not None:
        F.conv2d = pol
        torch.nn.functional.conv2d = pol
    try:
        with torch.no_grad():
            out, flow = stn(small, return_flow=True, padding_mode=mode, input_img_for_sampling=x if full else None)
    finally:
        F.conv2d = REAL_CONV
    return out, flow, params['sim'], (pol.log if pol else None)


if __name__ == '__main__':
    torch.set_num_threads(8)
    import_reference()
    for ci in (0, 1):
        ref_out, ref_flow, ref_sim, _ = run(None, ci)
        r64_out, r64_flow, r64_sim, _ = run(None, ci, dtype=torch.float64)
        print(f'case {ci}: reference fp32 vs fp64: out {float((ref_out - r64_out).abs().max()):.2e} '
              f'flow {float((ref_flow - r64_flow).abs().max()):.2e} sim {float((ref_sim - r64_sim).abs().max()):.2e}')
        _, _, _, log = run(lambda i, xs, ws: 2, ci, n=1)
        nsim = next(k for k, e in enumerate(log) if k > 0 and e[1][1] == 3)       # second 3-channel stem = flow stage
        if ci == 0:
            for e in log:
                print('   ', e, 'SIM' if e[0] < nsim else 'FLOW')
        policies = {
            'all bf16x3': lambda i, xs, ws: 2,
            'sim stage bf16x6, flow stage bf16x3': lambda i, xs, ws: 3 if i < nsim else 2,
            'sim stage <= 8^2 layers bf16x6': lambda i, xs, ws: 3 if (i < nsim and xs[-1] <= 9) else 2,
            'sim stage <= 32^2 layers bf16x6': lambda i, xs, ws: 3 if (i < nsim and xs[-1] <= 33) else 2,
            'sim stage >= 64^2 layers bf16x6': lambda i, xs, ws: 3 if (i < nsim and xs[-1] >= 63) else 2,
            'sim stage fp32 exact, flow stage bf16x3': lambda i, xs, ws: 0 if i < nsim else 2,
            'all bf16x6': lambda i, xs, ws: 3,
            'all fp16x3 (binary16 limbs, 3 products)': lambda i, xs, ws: 16,
            'sim stage fp16x3, flow stage bf16x3': lambda i, xs, ws: 16 if i < nsim else 2,
        }
        for name, fn in policies.items():
            out, flow, sim, _ = run(fn, ci)
            print(f'  {name:44s} out {float((out - ref_out).abs().max()):.2e} (vs fp64 {float((out - r64_out).abs().max()):.2e})'
                  f'  flow {float((flow - ref_flow).abs().max()):.2e}  sim params {float((sim - ref_sim).abs().max()):.2e}'
                  f' rel {float(((sim - ref_sim).abs() / ref_sim.abs().clamp_min(1e-3)).max()):.2e}')
