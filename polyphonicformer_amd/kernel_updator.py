"""KernelUpdator -- drop-in for polyphonic/funcs/kernel_updator.py:6-93 (registered in
TRANSFORMER_LAYER under the same name, same constructor kwargs, same parameter names)."""
import torch
import torch.nn as nn

from . import _lib, engine as E
from .bricks import build_norm_layer
from .registry import register_everywhere


class KernelUpdator(nn.Module):

    def __init__(self, in_channels=256, feat_channels=64, out_channels=None, input_feat_shape=3,
                 gate_sigmoid=True, gate_norm_act=False, activate_out=False,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN')):
        super().__init__()
        if not (in_channels == feat_channels == 256 and (out_channels in (None, 256))):
            raise NotImplementedError("libpolyhead: KernelUpdator with in=feat=out=256 channels only")
        if not gate_sigmoid or gate_norm_act or activate_out or act_cfg.get('type') != 'ReLU' \
                or norm_cfg.get('type') != 'LN':
            raise NotImplementedError("libpolyhead: the shipped KernelUpdator variant only "
                                      "(gate_sigmoid, LN, ReLU, no gate_norm_act / activate_out)")
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.out_channels_raw, self.out_channels = out_channels, out_channels or in_channels
        self.gate_sigmoid, self.gate_norm_act, self.activate_out = gate_sigmoid, gate_norm_act, activate_out
        self.input_feat_shape = [input_feat_shape] * 2 if isinstance(input_feat_shape, int) else input_feat_shape
        self.act_cfg, self.norm_cfg = act_cfg, norm_cfg
        self.num_params_in = self.num_params_out = feat_channels
        self.dynamic_layer = nn.Linear(in_channels, 2 * feat_channels)
        self.input_layer = nn.Linear(in_channels, 2 * feat_channels, 1)
        self.input_gate = nn.Linear(in_channels, feat_channels, 1)
        self.update_gate = nn.Linear(in_channels, feat_channels, 1)
        self.norm_in = build_norm_layer(norm_cfg, feat_channels)[1]
        self.norm_out = build_norm_layer(norm_cfg, feat_channels)[1]
        self.input_norm_in = build_norm_layer(norm_cfg, feat_channels)[1]
        self.input_norm_out = build_norm_layer(norm_cfg, feat_channels)[1]
        self.activation = nn.ReLU(inplace=True)
        self.fc_layer = nn.Linear(feat_channels, self.out_channels, 1)
        self.fc_norm = build_norm_layer(norm_cfg, self.out_channels)[1]
        self.precision = "fp32"
        self._pack = None

    def _standalone_pack(self, device):
        """a synthetic one-branch 'stage' (identity feat_transform, zero everything downstream) so the
        fused query kernel's PRE phase evaluates exactly this module."""
        prec = E.PREC[self.precision]
        key = (prec, str(device), _lib.param_versions(self))
        if self._pack is not None and self._pack[0] == key:
            return self._pack[1]
        sd = {}
        own = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        z = lambda *s: torch.zeros(*s)
        for ku, tr, sfx in (("kernel_update_conv.", "feat_transform.conv.", ""),
                            ("kernel_update_conv_depth.", "feat_depth_transform.conv.", "_depth")):
            for k, v in own.items():
                sd[ku + k] = v
            sd[tr + "weight"] = torch.eye(256).reshape(256, 256, 1, 1)
            sd[tr + "bias"] = z(256)
            sd["attention" + sfx + ".attn.in_proj_weight"], sd["attention" + sfx + ".attn.in_proj_bias"] = z(768, 256), z(768)
            sd["attention" + sfx + ".attn.out_proj.weight"], sd["attention" + sfx + ".attn.out_proj.bias"] = z(256, 256), z(256)
            for n in ("attention_norm", "ffn_norm"):
                sd[n + sfx + ".weight"], sd[n + sfx + ".bias"] = torch.ones(256), z(256)
            sd["ffn" + sfx + ".layers.0.0.weight"], sd["ffn" + sfx + ".layers.0.0.bias"] = z(256, 256), z(256)
            sd["ffn" + sfx + ".layers.1.weight"], sd["ffn" + sfx + ".layers.1.bias"] = z(256, 256), z(256)
        for n in ("cls_fcs", "mask_fcs", "depth_regs"):
            sd[n + ".0.weight"], sd[n + ".1.weight"], sd[n + ".1.bias"] = z(256, 256), torch.ones(256), z(256)
        sd["fc_cls.weight"], sd["fc_cls.bias"] = z(16, 256), z(16)
        for n in ("fc_mask", "fc_depth"):
            sd[n + ".weight"], sd[n + ".bias"] = z(256, 256), z(256)
        pack = E.StagePack(sd, "", 16, prec, device)
        self._pack = (key, pack)
        return pack

    def forward(self, update_feature, input_feature):
        """update_feature [..., 256] (pooled feature), input_feature [..., 256] (kernel), with
        K*K == 1 (conv_kernel_size = 1).  Returns [num_proposals, 1, 256] like the reference (:93)."""
        u = update_feature.reshape(-1, self.in_channels)
        n = u.shape[0]
        k = input_feature.reshape(n, -1, self.in_channels)
        if k.shape[1] != 1:
            raise NotImplementedError("libpolyhead: conv_kernel_size == 1 only")
        E._require_gpu(u, "update_feature")
        dev = u.device
        pack = self._standalone_pack(dev)
        out = torch.empty((n, 1, self.out_channels), dtype=torch.float32, device=dev)
        # rows are independent: process in frames of <= 256 query rows
        for s in range(0, n, 256):
            m = min(256, n - s)
            Npad = E.n_padded(m)
            partial = torch.zeros((1, 1, Npad, 512), dtype=torch.float32, device=dev)
            partial[0, 0, :m, :256] = u[s:s + m].float()
            bits = torch.zeros((1, Npad, 4), dtype=torch.int32, device=dev)       # zero pixel count
            kin = k[s:s + m, 0].float().contiguous()[None]
            ws = torch.empty((_lib.load().ph_query_workspace_bytes(1, m, pack.prec),), dtype=torch.uint8, device=dev)
            E.query_stage(partial, bits, kin, torch.zeros_like(kin), pack, m, 128, workspace=ws, phases=1)
            off = _lib.load().ph_query_workspace_updator_offset(1, m, pack.prec)
            o1 = ws[off:off + 2 * Npad * 256 * 4].view(torch.float32).reshape(2, Npad, 256)
            out[s:s + m, 0] = o1[0, :m]
        return out


register_everywhere(KernelUpdator, "transformer_layer")
