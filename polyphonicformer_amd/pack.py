"""Host-side weight packing for `ph_query_stage` (include/polyhead.h: ph_stage_layout).

One KernelUpdateHead stage (reference parameter names) becomes
  wb : int16 [P][plane]  -- bf16 bit patterns, every Linear as MFMA 16x16x32 B-fragments in
                            tile-major order: block (ct, ks) = 64 lanes x 8 bf16, lane = 16*g + j
                            holds W[16*ct + j][32*ks + 8*g .. +8]   (one contiguous 1 KiB block)
  wf : fp32 [..]         -- biases, LayerNorm affine, folded vectors
P = 1 (bf16) or 2 (hi, lo with W ~= hi + lo).

Folding (done in float64): `feat_transform` / `feat_depth_transform` are 1x1 convs
without norm/activation (kernel_update_head.py:124-140), so
    pooled(M, W x + b)      = pooled(M, x) W^T + count(M) b^T        -> folded into dynamic_layer
    k . (W x + b)           = (k W) x + k . b                        -> folded into fc_mask / fc_depth
which removes two full-map 256x256 GEMMs per stage from the device path.
"""
import numpy as np
import torch

from . import _lib

C = 256


def _bf16_planes(w64, planes):
    """float64 [..] -> list of int16 tensors (bf16 bits): hi, and lo = bf16(w - hi) if planes == 2"""
    w = w64.to(torch.float32)
    hi = w.to(torch.bfloat16)
    out = [hi.view(torch.int16)]
    if planes == 2:
        lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
        out.append(lo.view(torch.int16))
    return out


def pack_b_fragments(w):
    """[Nout][K] (Nout % 16 == 0, K % 32 == 0) -> flat tile-major fragment order (see module doc)."""
    nout, k = w.shape
    assert nout % 16 == 0 and k % 32 == 0, (nout, k)
    t = w.reshape(nout // 16, 16, k // 32, 4, 8)        # [ct][j][ks][g][e]
    return t.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)   # [ct][ks][g][j][e]


def pack_b32(w):
    """[Nout][K] (Nout % 32 == 0, K % 16 == 0) -> B fragments of the 32x32x16 MFMA, tile-major: one contiguous
    1 KiB block per (32-column tile, 16-deep k-step); inside a block lane l = (column l & 31, k-group l >> 5)
    holds 8 consecutive k (csrc/ph_neck.hip: k_conv_nhwc)."""
    nout, k = w.shape
    assert nout % 32 == 0 and k % 16 == 0, (nout, k)
    t = w.reshape(nout // 32, 32, k // 16, 2, 8)        # [ct][n][ks][g][e]
    return t.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)   # [ct][ks][g][n][e]


def _pad_rows(w, mult=16):
    r = (-w.shape[0]) % mult
    if r:
        w = torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], 0)
    return w


def pack_stage(sd, prefix, num_classes, prec):
    """sd: dict of CPU tensors with the reference's key names under `prefix`
    (e.g. 'mask_head.0.').  Returns (wb int16 [P][plane], wf float32, StageLayout)."""
    hybrid = prec == _lib.PH_PREC_QHYBRID
    planes = 2 if (prec == _lib.PH_PREC_SPLIT or hybrid) else 1
    g = lambda k: sd[prefix + k].detach().to("cpu", torch.float64)
    lay = _lib.StageLayout()
    wparts, vparts, wnames = [], [], []
    woff = voff = 0
    # hybrid grade: the matrices the POST kernel multiplies with (every one of them has a LayerNorm- or softmax-bounded
    # operand on the other side) are ONE fp16 plane, the PRE kernel's (pooled sums, gate products: unbounded) hi + lo bf16
    # -- except its last three calls, the attention in-projection (input: a LayerNorm output; results stored as fp16 anyway)
    POST = {"OUT", "FFN1", "FFN2", "H0A", "H0B", "CLS", "KERN", "QKV"}

    def add_w(br, name, mat):
        nonlocal woff
        mat = _pad_rows(mat)
        lay.w[br][_lib.W_IDX[name]] = woff
        wparts.append(mat)
        wnames.append(name)
        woff += mat.numel()

    def add_v(br, name, vec):
        nonlocal voff
        lay.v[br][_lib.V_IDX[name]] = voff
        vparts.append(vec.reshape(-1))
        voff += vec.numel()

    F = g("ffn.layers.0.0.weight").shape[0]
    assert g("ffn.layers.0.0.weight").shape[1] == C, "in_channels must be 256"
    assert F % 256 == 0
    for br, (ku, tr, sfx) in enumerate((("kernel_update_conv.", "feat_transform.conv.", ""),
                                        ("kernel_update_conv_depth.", "feat_depth_transform.conv.", "_depth"))):
        Wx = g(tr + "weight").reshape(C, C)          # [c_out][c_in]
        bx = g(tr + "bias")
        Wdyn = g(ku + "dynamic_layer.weight")
        add_w(br, "DYN", Wdyn @ Wx)
        add_v(br, "DYN_CNT", Wdyn @ bx)
        add_v(br, "DYN_B", g(ku + "dynamic_layer.bias"))
        add_w(br, "INP", g(ku + "input_layer.weight"))
        add_v(br, "INP_B", g(ku + "input_layer.bias"))
        add_w(br, "IG", g(ku + "input_gate.weight"))
        add_v(br, "IG_B", g(ku + "input_gate.bias"))
        add_w(br, "UG", g(ku + "update_gate.weight"))
        add_v(br, "UG_B", g(ku + "update_gate.bias"))
        for nm, key in (("LN_IG", "input_norm_in"), ("LN_UG", "norm_in"), ("LN_PO", "norm_out"),
                        ("LN_IO", "input_norm_out"), ("LN_FC", "fc_norm")):
            add_v(br, nm + "_G", g(ku + key + ".weight"))
            add_v(br, nm + "_B", g(ku + key + ".bias"))
        add_w(br, "FC", g(ku + "fc_layer.weight"))
        add_v(br, "FC_B", g(ku + "fc_layer.bias"))
        at = "attention" + sfx + ".attn."
        add_w(br, "QKV", g(at + "in_proj_weight"))
        add_v(br, "QKV_B", g(at + "in_proj_bias"))
        add_w(br, "OUT", g(at + "out_proj.weight"))
        add_v(br, "OUT_B", g(at + "out_proj.bias"))
        add_v(br, "LN_ATT_G", g("attention_norm" + sfx + ".weight"))
        add_v(br, "LN_ATT_B", g("attention_norm" + sfx + ".bias"))
        ff = "ffn" + sfx + ".layers."
        add_w(br, "FFN1", g(ff + "0.0.weight"))
        add_v(br, "FFN1_B", g(ff + "0.0.bias"))
        add_w(br, "FFN2", g(ff + "1.weight"))
        add_v(br, "FFN2_B", g(ff + "1.bias"))
        add_v(br, "LN_FFN_G", g("ffn_norm" + sfx + ".weight"))
        add_v(br, "LN_FFN_B", g("ffn_norm" + sfx + ".bias"))
        if br == 0:
            add_w(br, "H0A", g("cls_fcs.0.weight"))
            add_v(br, "LN_H0A_G", g("cls_fcs.1.weight"))
            add_v(br, "LN_H0A_B", g("cls_fcs.1.bias"))
            add_w(br, "H0B", g("mask_fcs.0.weight"))
            add_v(br, "LN_H0B_G", g("mask_fcs.1.weight"))
            add_v(br, "LN_H0B_B", g("mask_fcs.1.bias"))
            Wc, bc = g("fc_cls.weight"), g("fc_cls.bias")
            assert Wc.shape[0] == num_classes
            add_w(br, "CLS", Wc)
            add_v(br, "CLS_B", torch.cat([bc, bc.new_zeros((-len(bc)) % 16)]))
            Wk, bk = g("fc_mask.weight"), g("fc_mask.bias")
        else:
            add_w(br, "H0A", g("depth_regs.0.weight"))
            add_v(br, "LN_H0A_G", g("depth_regs.1.weight"))
            add_v(br, "LN_H0A_B", g("depth_regs.1.bias"))
            Wk, bk = g("fc_depth.weight"), g("fc_depth.bias")
        # kern[n][c] = sum_c' (m1 Wk^T + bk)[n][c'] Wx[c'][c];  kbias[n] = (m1 Wk^T + bk)[n] . bx
        Wfold = torch.cat([Wx.t() @ Wk, (bx @ Wk)[None]], 0)            # [257][256]
        bfold = torch.cat([Wx.t() @ bk, (bx @ bk)[None]], 0)            # [257]
        add_w(br, "KERN", _pad_rows(Wfold))                             # -> [272][256]
        add_v(br, "KERN_B", torch.cat([bfold, bfold.new_zeros(272 - 257)]))

    if hybrid:
        hi, lo = [], []
        for nm, m in zip(wnames, wparts):
            fr = pack_b_fragments(_pad_rows(m))
            if nm in POST:
                h16 = fr.to(torch.float32).to(torch.float16).view(torch.int16)
                hi.append(h16)
                lo.append(torch.zeros_like(h16))
            else:
                ph, pl = _bf16_planes(fr, 2)
                hi.append(ph)
                lo.append(pl)
        frag = torch.cat([pack_b_fragments(_pad_rows(m)) for m in wparts])
        wb = torch.stack([torch.cat(hi), torch.cat(lo)], 0).contiguous()
    else:
        frag = torch.cat([pack_b_fragments(_pad_rows(m)) for m in wparts])
        wb = torch.stack(_bf16_planes(frag, planes), 0).contiguous()
    wf = torch.cat(vparts).to(torch.float32).contiguous()
    lay.wb_plane_elems = frag.numel()
    lay.ffn_dim = F
    lay.num_classes = num_classes
    return wb, wf, lay


def unpack_b_fragments(flat, nout, k):
    """inverse of pack_b_fragments (tests)."""
    t = flat.reshape(nout // 16, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4)
    return t.reshape(nout, k)
