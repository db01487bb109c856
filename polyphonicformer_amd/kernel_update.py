"""KernelUpdateIterHead -- drop-in for polyphonic/kernel_update.py:13-535 (inference side).
Same registry name, constructor kwargs, `mask_head.{s}.*` state_dict keys and method signatures."""
import torch
import torch.nn as nn

from . import _lib, engine as E
from .registry import ConfigDict, build_head, deep_cfg, register_everywhere


class KernelUpdateIterHead(nn.Module):

    def __init__(self, num_stages=6, recursive=False, assign_stages=5, stage_loss_weights=(1, 1, 1, 1, 1, 1),
                 do_panoptic=False, proposal_feature_channel=256, merge_cls_scores=False, post_assign=False,
                 hard_target=False, merge_joint=True, num_proposals=100, num_thing_classes=80,
                 num_stuff_classes=53, mask_assign_stride=4, ignore_label=255, tracking=False,
                 mask_head=None, mask_out_stride=4, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert mask_head is not None
        assert len(stage_loss_weights) == num_stages
        self.num_stages, self.stage_loss_weights = num_stages, stage_loss_weights
        self.proposal_feature_channel, self.merge_cls_scores = proposal_feature_channel, merge_cls_scores
        self.recursive, self.post_assign, self.mask_out_stride = recursive, post_assign, mask_out_stride
        self.hard_target, self.assign_stages, self.do_panoptic = hard_target, assign_stages, do_panoptic
        self.merge_joint, self.num_thing_classes, self.num_stuff_classes = merge_joint, num_thing_classes, num_stuff_classes
        self.mask_assign_stride, self.num_proposals, self.ignore_label = mask_assign_stride, num_proposals, ignore_label
        self.tracking = tracking
        self.train_cfg = train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        self.init_mask_head(None, mask_head)
        self.init_assigner_sampler()
        self.precision = "fp32"           # a key of engine.MODES: "fp32" (parity grade), "mixed", "fp16", "bf16" (fast)
        self.output_dtype = torch.float32
        self._plans = {}

    def init_mask_head(self, mask_roi_extractor, mask_head):
        """kernel_update.py:108-123; every stage gets its own copy of the config dict."""
        self.mask_head = nn.ModuleList()
        if not isinstance(mask_head, list):
            mask_head = [mask_head for _ in range(self.num_stages)]
        assert len(mask_head) == self.num_stages
        for head in mask_head:
            self.mask_head.append(build_head(deep_cfg(head)))
        if self.recursive:
            for i in range(self.num_stages):
                self.mask_head[i] = self.mask_head[0]

    def init_assigner_sampler(self):
        """kernel_update.py:88-102: one assigner and one sampler per stage, built from `train_cfg` (a dict is replicated)"""
        import copy
        from . import assigner as A
        self.mask_assigner, self.mask_sampler = [], []
        if self.train_cfg is not None:
            if isinstance(self.train_cfg, dict):
                self.train_cfg = [copy.deepcopy(self.train_cfg) for _ in range(self.num_stages)]
            self.train_cfg = [ConfigDict(c) if isinstance(c, dict) and not isinstance(c, ConfigDict) else c for c in self.train_cfg]
            for c in self.train_cfg:
                self.mask_assigner.append(A.build_assigner(dict(c.assigner)))
                self.mask_sampler.append(A.build_sampler(dict(c.sampler)))

    def init_weights(self):
        for i in range(self.num_stages):
            self.mask_head[i].init_weights()

    def set_precision(self, precision, output_dtype=None):
        assert precision in E.MODES, f"precision must be one of {sorted(E.MODES)}"
        self.precision = precision
        for h in self.mask_head:
            h.precision = precision
        if output_dtype is not None:
            self.output_dtype = output_dtype
        self._plans.clear()
        return self

    # -- execution -------------------------------------------------------------------------------
    # Round 6: a frame's outputs do not depend on the frames that share its call (engine.DecodePlan `frame_invariant`): the module
    # API's default.  False = launch geometry tuned to the batch (bench.py's throughput legs).
    frame_invariant = True

    def _plan(self, B, N, H, W, device):
        packs = [h.stage_pack(device, self.precision) for h in self.mask_head]
        key = (B, N, H, W, self.precision, self.output_dtype, str(device), tuple(id(p) for p in packs), bool(self.frame_invariant), E.plan_env_key())
        plan = self._plans.get(key)
        if plan is None:
            self._plans.clear()
            plan = E.DecodePlan(packs, B, N, H, W, E.MODES[self.precision], self.output_dtype, device, frame_invariant=bool(self.frame_invariant))
            self._plans[key] = plan
        return plan

    def _mask_forward(self, stage, x, object_feats, mask_preds, img_metas, depth_preds, depth_proposal, depth_feats):
        """kernel_update.py:125-157"""
        head = self.mask_head[stage]
        cls_score, mask_preds, object_feats, depth_preds, depth_proposal = head(
            x, object_feats, mask_preds, img_metas=img_metas, depth_preds=depth_preds,
            depth_proposal=depth_proposal, depth_feats=depth_feats)
        if head.mask_upsample_stride > 1 and (stage == self.num_stages - 1 or self.training):
            if head.mask_upsample_stride != 2:
                raise NotImplementedError("libpolyhead: mask_upsample_stride must be 1 or 2")
            scaled_mask_preds, scaled_depth_preds = E.upsample2x(mask_preds), E.upsample2x(depth_preds)
        else:
            scaled_mask_preds, scaled_depth_preds = mask_preds, depth_preds
        return dict(cls_score=cls_score, mask_preds=mask_preds, scaled_mask_preds=scaled_mask_preds,
                    object_feats=object_feats, scaled_depth_preds=scaled_depth_preds, depth_preds=depth_preds,
                    depth_proposal=depth_proposal)

    def _decode(self, x, proposal_feats, mask_preds, depth_feats, depth_proposal):
        B, N = proposal_feats.shape[:2]
        H, W = x.shape[-2:]
        E._require_gpu(x, "x")
        if self.mask_head[0].mask_upsample_stride != 2:
            raise NotImplementedError("libpolyhead: mask_upsample_stride == 2 (the shipped config)")
        if not self.mask_head[-1].loss_cls.use_sigmoid:
            raise NotImplementedError("libpolyhead: sigmoid classification only (the shipped config)")
        plan = self._plan(B, N, H, W, x.device)
        plan.renew_outputs()         # results are the caller's: an earlier call's tensors are never overwritten
        ho = getattr(x, "_ph_handoff", None)
        same_call = (ho is not None and ho["mask_preds"] is mask_preds and ho["depth_feats"] is depth_feats
                     and tuple(ho["bits"].shape) == tuple(plan.bits.shape))
        if same_call and ho["prec"] == plan.prec and plan.mode.name in ("bf16", "fp32", "fp16") \
                and tuple(ho["xp"].shape) == tuple(plan.xp.shape):
            # inputs come straight from this package's KernelHead: its planes (same format) and mask bits are reused
            plan.run_from_planes(ho["xp"], ho["dp"], ho["bits"], proposal_feats, depth_proposal)
        elif same_call and ho["prec"] == _lib.PH_PREC_SPLIT and plan.mode.feat == _lib.PH_PREC_BF16 \
                and tuple(ho["xp"].shape[1:]) == tuple(plan.xp.shape[1:]):
            # KernelHead at the parity grade (hi + lo bf16 planes) in front of a mode that reads ONE bf16 plane ('mixed',
            # 'mixed16', 'bf16'): the hi plane IS the bf16 rounding of x_feats the ingest pass would produce -- adopt it
            plan.run_from_planes(ho["xp"][:1], ho["dp"][:1], ho["bits"], proposal_feats, depth_proposal)
        else:
            plan.set_inputs(x, depth_feats, proposal_feats, depth_proposal, mask_preds)
            plan.run()
        return plan.outputs()

    def simple_test_mask_preds(self, x, proposal_feats, mask_preds, cls_score, img_metas, depth_preds=None,
                               depth_feats=None, depth_proposal=None, imgs_whwh=None, rescale=False):
        """kernel_update.py:356-401 -- returns (object_feats, cls_score, mask_preds, scaled_mask_preds)."""
        o = self._decode(x, proposal_feats, mask_preds, depth_feats, depth_proposal)
        B, N = proposal_feats.shape[:2]
        return o["obj"].reshape(B, N, 256, 1, 1), o["cls"], o["mask"], o["mask_up"]

    def simple_test(self, x, proposal_feats, mask_preds, cls_score, img_metas, depth_preds=None, depth_feats=None,
                    depth_proposal=None, imgs_whwh=None, aspp_semantic=None, rescale=False, semantic_input=None):
        """kernel_update.py:282-354"""
        if aspp_semantic is not None:
            raise NotImplementedError("semantic_aspp is not part of the shipped configs")
        if not self.do_panoptic:
            raise NotImplementedError        # as the reference (:353)
        o = self._decode(x, proposal_feats, mask_preds, depth_feats, depth_proposal)
        depth_init = E.upsample2x(depth_preds.float().contiguous())            # :302-307
        from .panoptic import get_panoptic
        results = []
        for b in range(len(img_metas)):
            results.append(get_panoptic(self, o["cls"][b], o["mask_up"][b], o["depth_up"][b], depth_init[b],
                                        img_metas[b]))
        return results

    def aug_test(self, features, proposal_list, img_metas, rescale=False):
        raise NotImplementedError('SparseMask does not support `aug_test`')

    def forward_train(self, x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, gt_depth=None,
                      depth_preds=None, depth_feats=None, depth_proposal=None, gt_bboxes_ignore=None, imgs_whwh=None,
                      gt_bboxes=None, gt_sem_seg=None, gt_sem_cls=None, with_grads=False):
        """kernel_update.py:159-280: every stage's predictions, the Hungarian assignment on the previous stage's detached
        predictions (`assigner.py`, ph_match_sums), pseudo sampling, `get_targets` and the stage's losses -- the dict of
        `s{stage}_{loss}` the reference returns (with `tracking=True` also object_feats, cls_score, mask_preds,
        scaled_mask_preds).  It TRAINS: `train.roi_forward_train` runs the stages as differentiable libpolyhead operations
        and attaches the 'loss' entries to one autograd node per stage, so that `sum(losses).backward()` -- what mmdet's
        `_parse_losses` + `backward()` do (mmdet/models/detectors/base.py:176-199) -- leaves the gradient of the objective
        on every parameter and on whatever of x / proposal_feats / depth_feats / ... is on a graph (the tensors
        `KernelHead.forward_train` returns are).  `with_grads=True` adds losses['_grads']: per stage d(stage losses) /
        d(its cls_score, scaled mask and depth predictions)."""
        from . import train as T
        if self.post_assign:
            raise NotImplementedError                       # as the reference (:222-223)
        if not self.mask_assigner:
            raise ValueError("forward_train needs train_cfg (assigner / sampler per stage)")
        if cls_score is not None:
            raise NotImplementedError("libpolyhead: the shipped KernelHead hands cls_scores=None to the roi head (kernel_head.py:291)")
        E._require_gpu(x, "x")
        B, N = proposal_feats.shape[:2]
        with torch.enable_grad():          # also around the reshapes: a view made with autograd off is cut from the graph
            k = proposal_feats.reshape(B, N, -1)
            q = depth_proposal.reshape(depth_proposal.shape[0], depth_proposal.shape[1], -1).expand(B, N, -1)
            losses, last = T.roi_forward_train(self, x.float(), depth_feats.float(), k.float(), mask_preds.float(), q.float(),
                                               depth_preds.float(), img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth,
                                               want_grads=with_grads)
            obj, cls, m, smask = last
            obj = obj.reshape(B, N, -1, 1, 1)
        if not self.tracking:
            return losses
        return losses, obj, cls, m, smask


register_everywhere(KernelUpdateIterHead)
