"""ResidualCoder — SECOND's anchor-relative box code (semantics of reference pcdet/utils/box_coder_utils.py:5-79)."""
import torch


class ResidualCoder(object):
    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        self.code_size = code_size + (1 if encode_angle_by_sincos else 0)
        self.encode_angle_by_sincos = encode_angle_by_sincos

    def encode_torch(self, boxes, anchors):
        """boxes, anchors (..., 7+C) [x,y,z,dx,dy,dz,heading,...] -> residual code (..., code_size).
        Sizes are clamped to >= 1e-5 (on copies: the reference clamps its arguments in place)."""
        a_xyz, a_sz, a_r, a_rest = anchors[..., 0:3], anchors[..., 3:6].clamp_min(1e-5), anchors[..., 6:7], anchors[..., 7:]
        g_xyz, g_sz, g_r, g_rest = boxes[..., 0:3], boxes[..., 3:6].clamp_min(1e-5), boxes[..., 6:7], boxes[..., 7:]
        diag = torch.sqrt(a_sz[..., 0:1] ** 2 + a_sz[..., 1:2] ** 2)
        t_xy = (g_xyz[..., 0:2] - a_xyz[..., 0:2]) / diag
        t_z = (g_xyz[..., 2:3] - a_xyz[..., 2:3]) / a_sz[..., 2:3]
        t_sz = torch.log(g_sz / a_sz)
        if self.encode_angle_by_sincos:
            t_r = torch.cat([torch.cos(g_r) - torch.cos(a_r), torch.sin(g_r) - torch.sin(a_r)], dim=-1)
        else:
            t_r = g_r - a_r
        return torch.cat([t_xy, t_z, t_sz, t_r, g_rest - a_rest], dim=-1)

    def decode_torch(self, box_encodings, anchors):
        """inverse of encode_torch; (B,N,7+C) or (N,7+C)."""
        a_xyz, a_sz, a_r, a_rest = anchors[..., 0:3], anchors[..., 3:6], anchors[..., 6:7], anchors[..., 7:]
        t = box_encodings
        diag = torch.sqrt(a_sz[..., 0:1] ** 2 + a_sz[..., 1:2] ** 2)
        g_xy = t[..., 0:2] * diag + a_xyz[..., 0:2]
        g_z = t[..., 2:3] * a_sz[..., 2:3] + a_xyz[..., 2:3]
        g_sz = torch.exp(t[..., 3:6]) * a_sz
        if self.encode_angle_by_sincos:
            g_r = torch.atan2(t[..., 7:8] + torch.sin(a_r), t[..., 6:7] + torch.cos(a_r))
            rest = t[..., 8:]
        else:
            g_r = t[..., 6:7] + a_r
            rest = t[..., 7:]
        return torch.cat([g_xy, g_z, g_sz, g_r, rest + a_rest], dim=-1)
