"""The HIP path behind the CPU oracle's numpy interface (oracle/oracle.py), so that the closed-form / named-threshold
cases of tests/test_oracle_known_answers.py run against BOTH implementations: numpy in, numpy out, every call through
the product's Python layer and the C ABI.  GPU only."""
import ctypes as C

import numpy as np
import torch


def _d(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


class HipBackend:
    name = "hip"

    def __init__(self):
        import mtgs_amd
        from mtgs_amd import wrapper
        self.gs, self.w = mtgs_amd, wrapper

    def sh_fwd(self, degree, dirs, coeffs, masks=None):
        m = None if masks is None else torch.as_tensor(np.ascontiguousarray(masks)).bool().cuda()
        return self.gs.spherical_harmonics(degree, _d(dirs), _d(coeffs), masks=m).cpu().numpy()

    def project_fwd(self, means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0,
                    calc_compensations=False):
        out = self.w.fully_fused_projection(_d(means), None, _d(quats), _d(scales), _d(viewmats), _d(Ks), W, H, eps2d=eps2d,
                                            near_plane=near, far_plane=far, radius_clip=radius_clip,
                                            calc_compensations=calc_compensations)
        return tuple(None if t is None else t.cpu().numpy() for t in out)

    def isect_tiles(self, means2d, radii, depths, tile_size, tw, th, sort=True):
        tpg, ids, flat = self.w.isect_tiles(_d(means2d), _d(radii, torch.int32), _d(depths), tile_size, tw, th, sort=sort)
        return tpg.cpu().numpy(), ids.cpu().numpy(), flat.cpu().numpy()

    def isect_offset_encode(self, isect_ids, Cc, tw, th):
        return self.w.isect_offset_encode(_d(isect_ids, torch.int64), Cc, tw, th).cpu().numpy()

    def sort_pairs(self, keys, vals, key_bits):
        from mtgs_amd._lib import call, ptr
        k, v = _d(keys, torch.int64), _d(vals, torch.int32)
        ko, vo = torch.empty_like(k), torch.empty_like(v)
        n = C.c_size_t(0)
        call("mtgs_sort_workspace_bytes", k.numel(), C.byref(n))
        ws = torch.empty(n.value, dtype=torch.uint8, device="cuda")
        call("mtgs_sort_pairs", k.numel(), key_bits, ptr(k), ptr(v), ptr(ko), ptr(vo), ptr(ws), n.value,
             torch.cuda.current_stream().cuda_stream)
        return ko.cpu().numpy(), vo.cpu().numpy()

    def rasterization(self, means, quats, scales, opacities, colors, viewmats, Ks, width, height, **kw):
        bg = kw.pop("backgrounds", None)
        render, alpha, info = self.gs.rasterization(_d(means), _d(quats), _d(scales), _d(opacities), _d(colors), _d(viewmats),
                                                    _d(Ks), width, height, packed=False,
                                                    backgrounds=None if bg is None else _d(bg), **kw)
        meta = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in info.items()}
        return render.detach().cpu().numpy(), alpha.detach().cpu().numpy(), meta

    def grads(self, means, quats, scales, opacities, colors, viewmats, Ks, width, height, Gc, Ga, **kw):
        """(means2d.grad, means2d.absgrad) of L = sum(render Gc) + sum(alpha Ga), absgrad=True."""
        P = [_d(t).requires_grad_(True) for t in (means, quats, scales, opacities, colors)]
        render, alpha, info = self.gs.rasterization(*P, _d(viewmats), _d(Ks), width, height, packed=False, absgrad=True, **kw)
        info["means2d"].retain_grad()
        torch.autograd.backward([render, alpha], [_d(Gc), _d(Ga)])
        return info["means2d"].grad.cpu().numpy(), info["means2d"].absgrad.cpu().numpy(), alpha.detach().cpu().numpy()
