"""KernelUpdateHead -- drop-in for polyphonic/kernel_update_head.py:18-353,593-626 (inference).
Same registry name, constructor kwargs, attribute names and state_dict keys; the arithmetic of
`forward` runs in libpolyhead (pool -> query stage -> dynamic conv)."""
import torch
import torch.nn as nn

from . import _lib, engine as E
from .bricks import (ConvModuleParams, FFNParams, MultiheadAttentionParams, bias_init_with_prob,
                     build_norm_layer)
from .registry import build_loss, build_transformer_layer, register_everywhere
from . import losses as _losses          # registers FocalLoss / CrossEntropyLoss / DiceLoss / DepthLoss under the reference's names


class KernelUpdateHead(nn.Module):

    def __init__(self, num_classes=80, num_thing_classes=80, num_stuff_classes=53, num_ffn_fcs=2, num_heads=8,
                 num_cls_fcs=1, num_mask_fcs=3, feedforward_channels=2048, in_channels=256, out_channels=256,
                 dropout=0.0, mask_thr=0.5, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_act_cfg=dict(type='ReLU', inplace=True), conv_kernel_size=3, feat_transform_cfg=None,
                 hard_mask_thr=0.5, kernel_init=False, with_ffn=True, mask_out_stride=4, relative_coors=False,
                 relative_coors_off=False, feat_gather_stride=1, mask_transform_stride=1, mask_upsample_stride=1,
                 mask_assign_stride=4, ignore_label=255,
                 kernel_updator_cfg=dict(type='DynamicConv', in_channels=256, feat_channels=64, out_channels=256,
                                         input_feat_shape=1, act_cfg=dict(type='ReLU', inplace=True),
                                         norm_cfg=dict(type='LN')),
                 loss_rank=None, loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0),
                 loss_dice=dict(type='DiceLoss', loss_weight=3.0),
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
                 loss_depth=dict(type='DepthLoss', loss_weight=1.0, act=True, si_weight=1.0, sq_rel_weight=1.0,
                                 abs_rel_weight=1.0),
                 depth_act_mode='monodepth'):
        super().__init__()
        unsupported = []
        if conv_kernel_size != 1: unsupported.append("conv_kernel_size != 1")
        if in_channels != 256 or out_channels != 256: unsupported.append("channels != 256")
        if num_heads != 8: unsupported.append("num_heads != 8")
        if num_ffn_fcs != 2 or not with_ffn: unsupported.append("FFN layout")
        if num_cls_fcs != 1 or num_mask_fcs != 1: unsupported.append("num_cls_fcs / num_mask_fcs != 1")
        if feat_transform_cfg is None: unsupported.append("feat_transform_cfg=None")
        if feat_gather_stride != 1 or mask_transform_stride != 1: unsupported.append("gather/transform stride")
        if dropout != 0.0: unsupported.append("dropout")
        if hard_mask_thr != 0.5: unsupported.append("hard_mask_thr != 0.5")
        if feedforward_channels % 256: unsupported.append("feedforward_channels % 256")
        if act_cfg.get('type') != 'ReLU': unsupported.append("act_cfg")
        if unsupported:
            raise NotImplementedError("libpolyhead implements the shipped KernelUpdateHead configuration "
                                      "(configs/_base_/models/polyphonic_former.py:111-165); got: "
                                      + ", ".join(unsupported))
        self.num_classes = num_classes
        self.loss_cls, self.loss_mask = build_loss(loss_cls), build_loss(loss_mask)
        self.loss_dice, self.loss_depth = build_loss(loss_dice), build_loss(loss_depth)
        self.loss_rank = build_loss(loss_rank) if loss_rank is not None else None
        self.in_channels, self.out_channels = in_channels, out_channels
        self.mask_thr, self.fp16_enabled, self.dropout = mask_thr, False, dropout
        self.num_heads, self.hard_mask_thr, self.kernel_init, self.with_ffn = num_heads, hard_mask_thr, kernel_init, with_ffn
        self.mask_out_stride, self.relative_coors, self.relative_coors_off = mask_out_stride, relative_coors, relative_coors_off
        self.conv_kernel_size, self.feat_gather_stride = conv_kernel_size, feat_gather_stride
        self.mask_transform_stride, self.mask_upsample_stride = mask_transform_stride, mask_upsample_stride
        self.num_thing_classes, self.num_stuff_classes = num_thing_classes, num_stuff_classes
        self.mask_assign_stride, self.ignore_label = mask_assign_stride, ignore_label

        self.attention = MultiheadAttentionParams(in_channels, num_heads, dropout)
        self.attention_depth = MultiheadAttentionParams(in_channels, num_heads, dropout)
        self.attention_norm = build_norm_layer(dict(type='LN'), in_channels)[1]
        self.attention_norm_depth = build_norm_layer(dict(type='LN'), in_channels)[1]
        self.kernel_update_conv = build_transformer_layer(kernel_updator_cfg)
        self.kernel_update_conv_depth = build_transformer_layer(kernel_updator_cfg)
        ft = dict(feat_transform_cfg)                    # the reference pops from the caller's dict (:125); we copy
        kernel_size = ft.pop('kernel_size', 1)
        if ft.get('act_cfg', 'x') is not None or ft.get('norm_cfg') is not None:
            raise NotImplementedError("feat_transform must be a bare conv (act_cfg=None, no norm) to be folded")
        self.feat_transform = ConvModuleParams(in_channels, in_channels, kernel_size, norm_cfg=None, act_cfg=None)
        self.feat_depth_transform = ConvModuleParams(in_channels, in_channels, kernel_size, norm_cfg=None, act_cfg=None)
        self.ffn = FFNParams(in_channels, feedforward_channels, num_ffn_fcs, act_cfg=ffn_act_cfg, dropout=dropout)
        self.ffn_norm = build_norm_layer(dict(type='LN'), in_channels)[1]
        self.ffn_depth = FFNParams(in_channels, feedforward_channels, num_ffn_fcs, act_cfg=ffn_act_cfg, dropout=dropout)
        self.ffn_norm_depth = build_norm_layer(dict(type='LN'), in_channels)[1]
        self.cls_fcs = nn.ModuleList([nn.Linear(in_channels, in_channels, bias=False),
                                      build_norm_layer(dict(type='LN'), in_channels)[1], nn.ReLU(inplace=True)])
        self.fc_cls = nn.Linear(in_channels, num_classes if self.loss_cls.use_sigmoid else num_classes + 1)
        self.mask_fcs = nn.ModuleList([nn.Linear(in_channels, in_channels, bias=False),
                                       build_norm_layer(dict(type='LN'), in_channels)[1], nn.ReLU(inplace=True)])
        self.depth_regs = nn.ModuleList([nn.Linear(in_channels, in_channels, bias=False),
                                         build_norm_layer(dict(type='LN'), in_channels)[1]])
        self.fc_mask = nn.Linear(in_channels, out_channels)
        self.fc_depth = nn.Linear(in_channels, out_channels)
        self.depth_act_mode = depth_act_mode
        self.precision = "fp32"       # a key of engine.MODES
        self._packs = {}

    # -- reference API ----------------------------------------------------------------------------
    def init_weights(self):
        """kernel_update_head.py:193-210"""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        if self.loss_cls.use_sigmoid:
            nn.init.constant_(self.fc_cls.bias, bias_init_with_prob(0.01))
        if self.kernel_init:
            nn.init.normal_(self.fc_mask.weight, mean=0, std=0.01)

    def stage_pack(self, device, precision=None):
        prec = E.MODES[precision or self.precision].query
        ver = _lib.param_versions(self)
        key = (prec, str(device))
        hit = self._packs.get(key)
        if hit is None or hit[0] != ver:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._packs[key] = (ver, E.StagePack(sd, "", self.fc_cls.out_features, prec, device))
        return self._packs[key][1]

    def forward(self, x, proposal_feat, mask_preds, prev_cls_score=None, mask_shape=None, img_metas=None,
                depth_preds=None, depth_proposal=None, depth_feats=None):
        """kernel_update_head.py:212-353.  x, depth_feats [B,256,H,W]; proposal_feat, depth_proposal
        [B,N,256,1,1]; mask_preds [B,N,H,W].  `depth_preds` is accepted and unused, as in the reference."""
        B, N = proposal_feat.shape[:2]
        H, W = x.shape[-2:]
        if tuple(mask_preds.shape[-2:]) != (H, W) or mask_shape is not None:
            raise NotImplementedError("libpolyhead: mask_preds must already be at the feature resolution")
        E._require_gpu(x, "x")
        mode = E.MODES[self.precision]
        pack = self.stage_pack(x.device)
        HW = H * W
        if x.dtype == mode.feat_dtype and depth_feats.dtype == x.dtype and HW % 128 == 0:
            # 16-bit NCHW tensors of the mode's plane format are the planes themselves
            xp, dp = (t.contiguous().view(torch.int16).reshape(1, B, 256, HW) for t in (x, depth_feats))
        else:
            xp, dp = E.ingest(x, mode.feat), E.ingest(depth_feats, mode.feat)
        bits = E.binarize(mask_preds)
        partial = E.pool(xp, dp, bits, N, HW, mode.feat)
        k = proposal_feat.reshape(B, N, 256).float().contiguous()
        q = depth_proposal.reshape(B, N, 256).float().contiguous()      # materialises the expand view
        o = E.query_stage(partial, bits, k, q, pack, N, HW, kern_fmt=mode.kern_fmt)
        new_mask = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
        new_depth = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
        E.dynconv(xp, o["kern"], o["kbias"], 0, N, HW, mode.conv, logits_out=new_mask)
        E.dynconv(dp, o["kern"], o["kbias"], 1, N, HW, mode.conv, logits_out=new_depth)
        return (o["cls"], new_mask, o["obj"].reshape(B, N, 256, 1, 1), new_depth, o["dobj"].reshape(B, N, 256, 1, 1))

    def loss(self, object_feats, cls_score, mask_pred, depth_pred, labels, label_weights, mask_targets, mask_weights,
             depth_targets, depth_weights, imgs_whwh=None, reduction_override=None, with_grads=False, **kwargs):
        """kernel_update_head.py:355-441: the stage's losses from its predictions at the assign stride and the targets of
        `get_targets` -- `loss_depth`, `loss_cls`, `pos_acc`, `loss_rpn_mask`, `loss_rpn_dice`, `loss_rank` (the reference's
        keys).  `with_grads=True` additionally returns d(sum of the losses) / d(mask_pred, cls_score, depth_pred)
        (csrc/ph_loss.hip).  Values only: the differentiable form is `train.roi_forward_train` (what
        `KernelUpdateIterHead.forward_train` runs), which attaches these losses to the graph."""
        if reduction_override is not None:
            raise NotImplementedError("libpolyhead: reduction_override is not used by the reference's training loop")
        return _losses.stage_losses(self, cls_score, mask_pred, depth_pred, labels, label_weights, mask_targets, mask_weights,
                                    depth_targets, depth_weights, with_grads=with_grads)

    def _get_target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls,
                           pos_depth, neg_depth, gt_depth, gt_valid, cfg):
        """kernel_update_head.py:443-531"""
        return _losses.target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg,
                                     gt_sem_cls, pos_depth, neg_depth, gt_depth, gt_valid, cfg)

    def get_targets(self, sampling_results, gt_mask, gt_labels, rcnn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None,
                    gt_depth=None):
        """kernel_update_head.py:533-591"""
        return _losses.get_targets(self, sampling_results, gt_mask, gt_labels, rcnn_train_cfg, concat, gt_sem_seg, gt_sem_cls,
                                   gt_depth)


register_everywhere(KernelUpdateHead)
