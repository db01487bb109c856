"""KernelHead -- drop-in for polyphonic/kernel_head.py:11-706.
Same registry name, constructor kwargs, attribute and state_dict names.  Everything after
`localization_fpn(img)` (kernel_head.py:243) runs in libpolyhead; `forward_train` trains (train.py: differentiable
libpolyhead operations, losses attached to one autograd node)."""
import torch
import torch.nn as nn

from . import _lib, engine as E
from .bricks import ConvModuleParams, bias_init_with_prob
from . import losses as _losses  # noqa: F401  registers FocalLoss / CrossEntropyLoss / DiceLoss / DepthLoss
from .registry import NECKS, ConfigDict, build_loss, build_neck, register_everywhere


class KernelHead(nn.Module):

    def __init__(self, num_proposals=100, num_classes=133, num_thing_classes=80, num_stuff_classes=53,
                 in_channels=256, out_channels=256, num_heads=8, num_cls_fcs=1, num_seg_convs=1, num_loc_convs=1,
                 att_dropout=False, localization_fpn=None, conv_kernel_size=1, norm_cfg=dict(type='GN', num_groups=32),
                 semantic_fpn=True, train_cfg=None, xavier_init_kernel=False, kernel_init_std=0.01, use_binary=False,
                 proposal_feats_with_obj=False, loss_mask=None, loss_seg=None, loss_cls=None, loss_dice=None,
                 loss_rank=None, loss_depth=None, feat_downsample_stride=1, feat_refine_stride=1, feat_refine=True,
                 conv_normal_init=False, mask_out_stride=4, hard_target=False, ignore_label=255, cat_stuff_mask=False,
                 with_depth=True, num_depth_convs=1, semantic_out_cfg=None, loss_semantic_seg=None, **kwargs):
        super().__init__()
        unsupported = []
        if in_channels != 256 or out_channels != 256: unsupported.append("channels != 256")
        if conv_kernel_size != 1: unsupported.append("conv_kernel_size != 1")
        if num_seg_convs != 1 or num_loc_convs != 1 or num_depth_convs != 1: unsupported.append("num_*_convs != 1")
        if not semantic_fpn or not with_depth: unsupported.append("semantic_fpn / with_depth off")
        if feat_downsample_stride > 1 and feat_refine: unsupported.append("feat_refine (3x3 downsample convs)")
        if not (use_binary and proposal_feats_with_obj): unsupported.append("use_binary / proposal_feats_with_obj off")
        if semantic_out_cfg: unsupported.append("semantic_out_cfg")
        if norm_cfg.get('type') != 'GN' or 256 % norm_cfg.get('num_groups', 32): unsupported.append("norm_cfg")
        if unsupported:
            raise NotImplementedError("libpolyhead implements the shipped KernelHead configuration "
                                      "(configs/_base_/models/polyphonic_former.py:30-97); got: " + ", ".join(unsupported))
        self.num_proposals, self.num_cls_fcs, self.train_cfg = num_proposals, num_cls_fcs, train_cfg
        self.in_channels, self.out_channels, self.num_classes = in_channels, out_channels, num_classes
        self.proposal_feats_with_obj, self.sampling = proposal_feats_with_obj, False
        if localization_fpn is None:
            self.localization_fpn = None
        elif isinstance(localization_fpn, nn.Module):
            self.localization_fpn = localization_fpn
        elif localization_fpn.get('type') in NECKS:
            self.localization_fpn = build_neck(localization_fpn)
        else:
            # an unknown neck type: `simple_test_rpn` then expects the neck's three output maps as `img`
            self.localization_fpn = None
        self.semantic_fpn, self.norm_cfg, self.num_heads, self.att_dropout = semantic_fpn, norm_cfg, num_heads, att_dropout
        self.mask_out_stride, self.hard_target, self.conv_kernel_size = mask_out_stride, hard_target, conv_kernel_size
        self.xavier_init_kernel, self.kernel_init_std = xavier_init_kernel, kernel_init_std
        self.feat_downsample_stride, self.feat_refine_stride = feat_downsample_stride, feat_refine_stride
        self.conv_normal_init, self.feat_refine = conv_normal_init, feat_refine
        self.num_loc_convs, self.num_seg_convs, self.use_binary = num_loc_convs, num_seg_convs, use_binary
        self.num_thing_classes, self.num_stuff_classes = num_thing_classes, num_stuff_classes
        self.ignore_label, self.cat_stuff_mask = ignore_label, cat_stuff_mask
        self.with_depth, self.num_depth_convs, self.semantic_out_cfg = with_depth, num_depth_convs, semantic_out_cfg
        bl = lambda c: build_loss(c) if c is not None else None
        self.loss_mask, self.loss_dice, self.loss_seg, self.loss_cls = bl(loss_mask), bl(loss_dice), bl(loss_seg), bl(loss_cls)
        self.loss_rank, self.loss_depth, self.loss_semantic_seg = bl(loss_rank), bl(loss_depth), bl(loss_semantic_seg)
        if self.loss_seg is None:
            raise ValueError("loss_seg is required when semantic_fpn=True (kernel_head.py:152)")
        # layers (kernel_head.py:142-211)
        self.init_kernels = nn.Conv2d(out_channels, num_proposals, 1, padding=0, bias=False)
        self.conv_seg = nn.Conv2d(out_channels, num_classes if self.loss_seg.use_sigmoid else num_classes + 1, 1)
        self.loc_convs = nn.ModuleList([ConvModuleParams(in_channels, out_channels, 1, norm_cfg=norm_cfg)])
        self.seg_convs = nn.ModuleList([ConvModuleParams(in_channels, out_channels, 1, norm_cfg=norm_cfg)])
        self.depth_convs = nn.ModuleList([ConvModuleParams(in_channels, out_channels, 1, norm_cfg=norm_cfg)])
        self.conv_direct_depth = nn.Conv2d(out_channels, 1, 1)
        self.semantic_aspp = None
        self.precision = "fp32"
        self.emit_fp32_features = True     # the reference API returns x_feats / depth_feats as fp32 NCHW tensors
        self.logit_dtype = torch.float32   # mask_preds / seg_preds / depth_pred; torch.float16 halves their bytes (one-pass form)
        self._pack, self._plans = None, {}
        self.assigner = self.sampler = None
        if self.train_cfg:                 # kernel_head.py:134-140
            from . import assigner as A
            if isinstance(self.train_cfg, dict) and not isinstance(self.train_cfg, ConfigDict):
                self.train_cfg = ConfigDict(self.train_cfg)
            self.assigner = A.build_assigner(dict(self.train_cfg.assigner))
            self.sampler = A.build_sampler(dict(self.train_cfg.get('sampler', None) or dict(type='MaskPseudoSampler')))

    def init_weights(self):
        """kernel_head.py:213-238"""
        if self.localization_fpn is not None and hasattr(self.localization_fpn, "init_weights"):
            self.localization_fpn.init_weights()
        if self.feat_downsample_stride > 1 and self.conv_normal_init:
            for conv in [self.loc_convs, self.seg_convs]:
                for m in conv.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.normal_(m.weight, 0, 0.01)
        if self.loss_seg.use_sigmoid:
            nn.init.normal_(self.conv_seg.weight, 0, 0.01)
            nn.init.constant_(self.conv_seg.bias, bias_init_with_prob(0.01))
        else:
            nn.init.normal_(self.conv_seg.weight, 0, 0.01)
        if self.xavier_init_kernel:
            nn.init.xavier_uniform_(self.init_kernels.weight)
        else:
            nn.init.normal_(self.init_kernels.weight, 0, self.kernel_init_std)

    def set_precision(self, precision):
        assert precision in E.PREC, f"precision must be one of {sorted(E.PREC)}"      # 'mixed' / 'fp16' run a1 at fp32 grade
        self.precision = precision
        self._pack, self._plans = None, {}
        if self.localization_fpn is not None and hasattr(self.localization_fpn, "set_precision"):
            self.localization_fpn.set_precision("fp32" if precision == "split" else precision)
        return self

    def _get_pack(self, device):
        prec = E.KHEAD_PREC[self.precision]
        own = {k: v for k, v in self.state_dict().items() if not k.startswith("localization_fpn.")}
        ver = _lib.param_versions(self)
        if self._pack is None or self._pack[0] != (prec, str(device), ver):
            self._pack = ((prec, str(device), ver),
                          E.KernelHeadPack(own, prec, device, self.norm_cfg.get('num_groups', 32)))
            self._plans = {}
        return self._pack[1]

    # Round 6: a frame's outputs do not depend on the frames that share its call (see KernelUpdateIterHead.frame_invariant)
    frame_invariant = True

    def _decode_init_proposals(self, img, img_metas, train_tracking=False):
        """kernel_head.py:240-347.  `img`: the FPN tuple when `localization_fpn` is a module, otherwise the three post-neck
        maps [B,256,H,W].  In training mode the stuff rows are not appended here (:329) -- `forward_train` does it (:444-451)."""
        neck = self.localization_fpn
        handoff = neck is not None and hasattr(neck, "forward_planes") and getattr(neck, "num_aux_convs", 0) == 2 \
            and E.KHEAD_PREC.get(getattr(neck, "precision", None)) == E.KHEAD_PREC[self.precision]   # codes: 'split' == 'fp32'
        if handoff:
            # this build's neck: its three maps come over as bf16 planes (half the bytes, no conversion pass in front of
            # the GEMMs; bit-identical in bf16 precision because the first use of the fp32 maps is that same rounding)
            E._require_gpu(img[0], "FPN levels")
            feats = neck.forward_planes(img)
            B, (H, W) = img[0].shape[0], tuple(img[1].shape[-2:])
            dev = feats[0].device
        else:
            feats = neck(img) if neck is not None else list(img)
            if not isinstance(feats, (list, tuple)) or len(feats) != 3:
                raise NotImplementedError("with_depth needs the neck's three maps (kernel_head.py:272-276)")
            E._require_gpu(feats[0], "localization feats")
            B, C, H, W = feats[0].shape
            dev = feats[0].device
        pack = self._get_pack(dev)
        cat_stuff = self.cat_stuff_mask and not self.training
        key = (B, H, W, cat_stuff, self.emit_fp32_features, self.logit_dtype, bool(self.frame_invariant))
        plan = self._plans.get(key)
        if plan is None:
            self._plans = {key: E.KernelHeadPlan(pack, B, H, W, self.num_thing_classes, self.num_classes, cat_stuff, dev,
                                                 want_f32=self.emit_fp32_features, logit_dtype=self.logit_dtype,
                                                 frame_invariant=bool(self.frame_invariant))}
            plan = self._plans[key]
        plan.renew_outputs()         # the 9-tuple (and the hand-off planes) belong to the caller from here on
        plan.set_inputs(list(feats) if handoff else [f.float() for f in feats])
        plan.run()
        N = plan.N
        proposal_feats = plan.proposal.reshape(B, N, 256, 1, 1)
        depth_proposal = pack.w_dd_f32[None].expand(B, N if cat_stuff else 1, 256, 1, 1)     # stride-0 view (:286-289,336)
        x_feats, depth_feats = plan.x_f32, plan.dfe_f32
        if x_feats is not None:
            # hand the bf16 planes / mask bits to KernelUpdateIterHead so that it can skip its ingest pass
            x_feats._ph_handoff = dict(xp=plan.xp, dp=plan.dp, bits=plan.bits, prec=pack.prec, mask_preds=plan.mask_preds,
                                       depth_feats=depth_feats)
        return (proposal_feats, x_feats, plan.mask_preds, None, plan.seg_preds, depth_feats, depth_proposal,
                plan.depth_pred, None)

    def simple_test_rpn(self, img, img_metas, train_tracking=False):
        """kernel_head.py:700-706"""
        return self._decode_init_proposals(img, img_metas, train_tracking)

    def forward_dummy(self, img, img_metas):
        return self._decode_init_proposals(img, img_metas)

    def forward_train(self, img, img_metas, gt_masks, gt_labels, gt_sem_seg=None, gt_sem_cls=None, gt_depth=None, with_grads=False):
        """kernel_head.py:349-454: the decode in training mode, the x`feat_downsample_stride` upsample of the mask / seg /
        depth predictions, the Hungarian assignment on the detached masks (`assigner.py`), pseudo sampling, `get_targets`,
        `loss` and `depth_dense`; returns the reference's 9-tuple, the stuff rows appended when `cat_stuff_mask`.
        `gt_depth`: [B, 1, H, W] at the scaled size, as the dataset pipeline delivers it.
        It TRAINS: the forward runs as differentiable libpolyhead operations (`train.rpn_forward_train`, hand-written
        backward of every map-sized product), the 'loss' entries of the returned dict are attached to one autograd node, and
        the tensors handed on to the roi head stay on the graph -- mmdet's `_parse_losses` + `backward()`
        (mmdet/models/detectors/base.py:176-199) work on the result unchanged.  `img`: the three post-neck maps (gradients
        flow into them when they require them) or, with `localization_fpn` set, the four FPN levels: the neck then runs its
        differentiable form (`train.neck_forward_train`) and is trained through this method like the reference's (round 5; the
        `frozen_neck_ok` escape hatch of rounds 3-4 is gone).  `with_grads=True` adds losses['_grads'] = d(sum of the 'loss' entries) / d(scaled mask, seg
        and direct depth predictions)."""
        from . import train as T
        if self.assigner is None:
            raise ValueError("forward_train needs train_cfg (assigner / sampler)")
        neck = self.localization_fpn
        with torch.enable_grad():      # the neck's differentiable form (SemanticFPNWrapper under autograd): it is trained from here
            feats = neck(img) if neck is not None else list(img)
        if not isinstance(feats, (list, tuple)) or len(feats) != 3:
            raise NotImplementedError("with_depth needs the neck's three maps (kernel_head.py:272-276)")
        for f in feats:
            E._require_gpu(f, "localization feats")
        feats = [f if f.dtype == torch.float32 and f.is_contiguous() else f.float().contiguous() for f in feats]
        with torch.enable_grad():
            losses, r = T.rpn_forward_train(self, feats, img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth,
                                            want_grads=with_grads)
            k, mask_preds, q = T.rpn_outputs(self, r)
            B, N = k.shape[:2]
            out = (losses, k.reshape(B, N, 256, 1, 1), r["x"], mask_preds, None, r["dfe"], q.reshape(B, N, 256, 1, 1), r["depth_pred"], None)
        return out

    def loss(self, mask_pred, cls_scores, seg_preds, depth_pred, proposal_feats, semantic_aspp_out, labels, label_weights,
             mask_targets, mask_weights, seg_targets, depth_targets, depth_weights, reduction_override=None, with_grads=False, **kwargs):
        """kernel_head.py:456-569 (`losses.rpn_losses`)"""
        from . import losses as Lo
        if cls_scores is not None or semantic_aspp_out is not None:
            raise NotImplementedError("libpolyhead: the shipped KernelHead has no cls_scores / semantic_aspp (kernel_head.py:291,318)")
        return Lo.rpn_losses(self, mask_pred, seg_preds, depth_pred, labels, label_weights, mask_targets, mask_weights, seg_targets,
                             depth_targets, depth_weights, with_grads=with_grads)

    def _get_target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls,
                           pos_depth, neg_depth, gt_depth, gt_valid, cfg):
        """kernel_head.py:571-647"""
        from . import losses as Lo
        return Lo.rpn_target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls,
                                    pos_depth, neg_depth, gt_depth, gt_valid, cfg)

    def get_targets(self, sampling_results, gt_mask, rpn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None, gt_depth=None):
        """kernel_head.py:649-698"""
        from . import losses as Lo
        return Lo.rpn_get_targets(self, sampling_results, gt_mask, rpn_train_cfg, concat, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                  gt_depth=gt_depth)


register_everywhere(KernelHead)
from . import semantic_fpn  # noqa: E402,F401  registers 'SemanticFPNWrapper' (the localization_fpn)
