"""UnVoxelOdomNetICP3: the network + self-supervised loss of the RSLO hot path
(reference: rslo/models/voxel_odom_net.py:47-816; call stacks SURVEY.md 3.3 / 3.4).

Same constructor kwargs, attribute names, state-dict keys and forward(example) -> dict contract as the
reference class, so second_builder / train_hdf5.py / evaluate.py can drive it unchanged.

MI355X re-design inside that contract:
  * all frames of all samples go through the sparse encoder as ONE batched SparseConvTensor (batch index
    = t * B + b): one rulebook chain and ~20 conv launches per step instead of per frame;
  * the loss is batched over frame pairs with static shapes (rslo/core/losses.py);
  * global_step is mirrored on the host (no .cpu() sync per query); the logging extras of the training
    forward stay on the GPU unless `cpu_extras` is True (the reference copies ~10 tensors to the host every
    step, voxel_odom_net.py:519-539).
"""
import contextlib
import os
import weakref
import time

import apex.amp as amp
import apex.parallel as _apex_parallel
import kornia
import numpy as np
import torch
import torchplus
from torch import nn
from torch.nn import functional as F

from rslo.core import losses
from rslo.data.dataset import _grid_geometry as _tq_map_geometry
from rslo.data.dataset import generate_pointwise_local_transformation_tch
from rslo.models import middle, odom_pred, voxel_encoder

_SIDE_STREAMS = {}
_HOST_LEAD = int(os.environ.get("RSLO_HOST_LEAD", "1"))
_GRAPH_COV = os.environ.get("RSLO_HEAD_GRAPH_COV", "before")      # covariance branch: "before" / "after" the replayed head
GATE_EVENTS = weakref.WeakKeyDictionary()      # network -> event where its latest training forward reached the loss
_LEAD_EVENTS = weakref.WeakKeyDictionary()   # network -> events recorded behind its recent training forwards
_LEAD_WAIT = [0.0, 0.0]  # wall seconds the issuing thread was held back, CPU seconds it spent in that wait (bench.py)

REGISTERED_NETWORK_CLASSES = {}


def register_voxelnet(cls, name=None):
    name = cls.__name__ if name is None else name
    assert name not in REGISTERED_NETWORK_CLASSES, f"exist class: {REGISTERED_NETWORK_CLASSES}"
    REGISTERED_NETWORK_CLASSES[name] = cls
    return cls


def get_voxelnet_class(name):
    assert name in REGISTERED_NETWORK_CLASSES, f"available class: {REGISTERED_NETWORK_CLASSES}"
    return REGISTERED_NETWORK_CLASSES[name]


def create_cycle_constraint_data(xs, cat_dim=1):
    """All (i < j) pairs of a list of [B, ...] tensors -> [x_i...], [x_j...] each [B*npairs, ...]."""
    assert len(xs) >= 2
    shape = xs[0].shape
    x1, x2 = [], []
    for i in range(len(xs)):
        for j in range(i + 1, len(xs)):
            x1.append(xs[i])
            x2.append(xs[j])
    if len(x1) == 1:      # two frames: the stack + reshape of one element is the tensor itself (no copy launches)
        return [x1[0].reshape(-1, *shape[1:]), x2[0].reshape(-1, *shape[1:])]
    return [torch.stack(x1, dim=cat_dim).reshape(-1, *shape[1:]), torch.stack(x2, dim=cat_dim).reshape(-1, *shape[1:])]


def pair_rows_meta(counts):
    """counts[t][b] = voxels of sample b in frame t (the frames hold their samples back to back) -> int32 [T + 1, B]:
    row t = first row of each sample inside frame t, row T = the rows every frame of the sample is truncated to (the
    shortest frame's, reference voxel_odom_net.py:646-651).  The operand of losses.pair_rows / pad_rows."""
    T_, B = len(counts), len(counts[0])
    offs = [[sum(counts[t][:b]) for b in range(B)] for t in range(T_)]
    lens = [min(counts[t][b] for t in range(T_)) for b in range(B)]
    return torch.tensor(offs + [lens], dtype=torch.int32)


def _detach_tree(x):
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (list, tuple)):
        return [_detach_tree(v) for v in x]
    if isinstance(x, dict):
        return {k: _detach_tree(v) for k, v in x.items()}
    return x


def _cpu_tree(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    if isinstance(x, (list, tuple)):
        return [_cpu_tree(v) for v in x]
    if isinstance(x, dict):
        return {k: _cpu_tree(v) for k, v in x.items()}
    return x


@register_voxelnet
class UnVoxelOdomNetICP3(nn.Module):
    def __init__(self, output_shape, pc_range=None, num_input_features=4, vfe_class_name="VoxelFeatureExtractor",
                 vfe_num_filters=[32, 128], with_distance=False, middle_class_name="SparseMiddleExtractor",
                 middle_num_input_features=-1, middle_num_filters_d1=[64], middle_num_filters_d2=[64, 64],
                 middle_use_leakyReLU=False, middle_bn_type="BN", middle_relu_type="ReLU",
                 odom_class_name="ResNetOdomPred", odom_num_input_features=-1, odom_layer_nums=[3, 5, 5],
                 odom_layer_strides=[2, 2, 2], odom_num_filters=[128, 128, 256], odom_upsample_strides=[1, 2, 4],
                 odom_num_upsample_filters=[256, 256, 256], odom_pooling_type="avg_pool", odom_pooling_size=1,
                 odom_cycle_constraint=False, odom_conv_type="official", odom_format="rx+t",
                 odom_pred_pyramid_motion=False, odom_use_deep_supervision=False, odom_dense_predict=False,
                 odom_use_loss_mask=True, odom_use_dynamic_mask=False, odom_use_corr=False, odom_dropout=0.2,
                 odom_bn_type="BN", odom_conf_type="linear", odom_use_SPGN=False, odom_use_leakyReLU=False,
                 odom_first_conv_groups=1, odom_use_se=False, odom_use_sa=False, vfe_use_norm=True,
                 odom_enc_use_norm=True, odom_use_svd=False, odom_dropout_input=False, odom_cubic_pred_height=5,
                 freeze_bn=False, freeze_bn_affine=False, freeze_bn_start_step=1e20, sync_bn=False, use_GN=False,
                 encode_background_as_zeros=True, rotation_loss=None, translation_loss=None,
                 pyramid_rotation_loss=None, pyramid_translation_loss=None, consistency_loss=None,
                 measure_time=False, voxel_generator=None, pyloss_exp_w_base=0.5, testing=False, icp_iter=2,
                 name="voxel_odom_net", **kwargs):
        super().__init__()
        self.name = name
        self.testing = testing
        self._encode_background_as_zeros = encode_background_as_zeros
        self._num_input_features = num_input_features
        self.voxel_generator = voxel_generator
        self._rotation_loss = rotation_loss
        self._translation_loss = translation_loss
        self._pyramid_rotation_loss = pyramid_rotation_loss
        self._pyramid_translation_loss = pyramid_translation_loss
        self._consistency_loss = consistency_loss
        self._conf_reg_loss = nn.MSELoss
        assert pyloss_exp_w_base > 0
        self._pyloss_exp_w_base = pyloss_exp_w_base
        assert icp_iter > 0, "The parameter of icp_iter should be larger than 0."
        self.icp_iter = icp_iter
        self.measure_time = measure_time
        self.cpu_extras = False
        self.fused_pyramid = True      # pyramid supervision through rslo_pyramid_l2_* (GPU tensors)
        # reductions + loss weights + total of all loss terms in one launch each way (rslo_loss_tail_*); the torch
        # formulation op by op (what the golden vectors of the reference pin) runs for CPU tensors / other configurations
        self.fused_loss_tail = os.environ.get("RSLO_FUSED_LOSS_TAIL", "1") != "0"

        self.voxel_feature_extractor = voxel_encoder.get_vfe_class(vfe_class_name)(
            num_input_features, vfe_use_norm, num_filters=vfe_num_filters, with_distance=with_distance,
            voxel_size=self.voxel_generator.voxel_size, pc_range=self.voxel_generator.point_cloud_range)
        self.middle_feature_extractor = middle.get_middle_class(middle_class_name)(
            output_shape, bn_type=middle_bn_type, use_GN=use_GN, sync_bn=sync_bn,
            use_leakyReLU=middle_use_leakyReLU, relu_type=middle_relu_type,
            num_input_features=middle_num_input_features, num_filters_down1=middle_num_filters_d1,
            num_filters_down2=middle_num_filters_d2)
        self.middle_feature_extractor_name = middle_class_name
        self.odom_predictor = odom_pred.get_odom_class(odom_class_name)(
            bn_type=odom_bn_type, enc_use_norm=odom_enc_use_norm, conv_type=odom_conv_type,
            layer_nums=odom_layer_nums, layer_strides=odom_layer_strides, num_filters=odom_num_filters,
            upsample_strides=odom_upsample_strides, num_upsample_filters=odom_num_upsample_filters,
            num_input_features=odom_num_input_features * 2, pooling_type=odom_pooling_type,
            pooling_size=odom_pooling_size, encode_background_as_zeros=True, use_groupnorm=use_GN, num_groups=32,
            dropout=odom_dropout, cycle_constraint=odom_cycle_constraint,
            pred_pyramid_motion=odom_pred_pyramid_motion, use_deep_supervision=odom_use_deep_supervision,
            use_loss_mask=odom_use_loss_mask, use_dynamic_mask=odom_use_dynamic_mask, odom_format=odom_format,
            point_cloud_range=pc_range, dense_predict=odom_dense_predict, use_correlation=odom_use_corr,
            conf_type=odom_conf_type, use_SPGN=odom_use_SPGN, use_leakyReLU=odom_use_leakyReLU,
            dropout_input=odom_dropout_input, first_conv_groups=odom_first_conv_groups, use_se=odom_use_se,
            use_sa=odom_use_sa, use_svd=odom_use_svd, cubic_pred_height=odom_cubic_pred_height,
            freeze_bn=freeze_bn, freeze_bn_affine=freeze_bn_affine, sync_bn=sync_bn, name="odomPred")

        self.freeze_bn = freeze_bn
        self.freeze_bn_affine = freeze_bn_affine
        self.freeze_bn_start_step = freeze_bn_start_step
        self.register_buffer("global_step", torch.LongTensor(1).zero_())
        self._step_cache = (None, 0)
        self.warm_flag = False
        self._time_dict, self._time_total_dict, self._time_count_dict = {}, {}, {}

    # ------------------------------------------------------------------ bookkeeping (driver-facing)
    def train(self, mode=True):
        super().train(mode)
        if self.freeze_bn and self.get_global_step() >= self.freeze_bn_start_step:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
                    if self.freeze_bn_affine:
                        for p in (m.weight, m.bias):
                            if p is not None:
                                p.requires_grad = False
        return self

    def update_global_step(self):
        ver, val = self._step_cache
        known = ver == self.global_step._version
        self.global_step += 1
        if known:      # keep the host mirror in step: no device read on the next get_global_step()
            self._step_cache = (self.global_step._version, val + 1)

    def get_global_step(self):
        ver, val = self._step_cache
        if ver != self.global_step._version:
            val = int(self.global_step.item())
            self._step_cache = (self.global_step._version, val)
        return val

    def clear_global_step(self):
        self.global_step.zero_()

    def clear_metrics(self):
        pass

    def start_timer(self, *names):
        if not self.measure_time:
            return
        torch.cuda.synchronize()
        for name in names:
            self._time_dict[name] = time.time()

    def end_timer(self, name):
        if not self.measure_time or name not in self._time_dict:
            return
        torch.cuda.synchronize()
        dt = time.time() - self._time_dict[name]
        self._time_count_dict[name] = self._time_count_dict.get(name, 0) + 1
        self._time_total_dict[name] = self._time_total_dict.get(name, 0.0) + dt

    def clear_timer(self):
        self._time_count_dict.clear()
        self._time_dict.clear()
        self._time_total_dict.clear()

    @contextlib.contextmanager
    def profiler(self):
        old = self.measure_time
        self.measure_time = True
        yield
        self.measure_time = old

    def get_avg_time_dict(self):
        return {k: v / max(1, self._time_count_dict[k]) for k, v in self._time_total_dict.items()}

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _merge_coords(coors, batch_size):
        merged = []
        for t, c in enumerate(coors):
            c = c.int()
            if t:
                c = c.clone()
                c[:, 0] += t * batch_size
            merged.append(c)
        return torch.cat(merged, 0)

    def plan_example(self, example):
        """Attach the encoder's rulebooks to `example` ahead of the step ("sparse_plan"): they depend on the voxel
        coordinates only.  network_forward picks the plan up instead of building it."""
        coors = example["coordinates"]
        B = example["num_voxels"][0].shape[0]
        example["sparse_plan"] = self.middle_feature_extractor.plan(self._merge_coords(coors, B), len(coors) * B,
                                                                    with_pairs=self.training)
        return example

    def network_forward(self, voxels, num_points, coors, batch_size, example):
        assert len(voxels) == len(num_points) == len(coors), "The lengths should be same."
        T = len(voxels)
        self.start_timer("voxel_feature_extractor")
        fm = example.get("_frame_major") if example is not None and voxels is example.get("voxels") else None
        if fm is not None:      # rslo_amd.plan.EncoderPlanner: the frames are views of one block -> one launch, no cat
            vf_all = self.voxel_feature_extractor(fm[0], fm[1], None)
            voxel_features = list(vf_all.split([v.shape[0] for v in voxels], dim=0))
        else:
            voxel_features = [self.voxel_feature_extractor(voxels[t], num_points[t], coors[t]) for t in range(T)]
            vf_all = None
        self.end_timer("voxel_feature_extractor")

        self.start_timer("middle forward")
        # one batched encoder pass: frame t of sample b gets batch index t * B + b
        plan = example.get("sparse_plan") if example is not None else None
        if plan is None:
            plan = self.middle_feature_extractor.plan(self._merge_coords(coors, batch_size), T * batch_size)
        feats_all = torch.cat(voxel_features, 0) if vf_all is None else vf_all
        two_streams = feats_all.is_cuda and os.environ.get("RSLO_COV_STREAM", "1") != "0"
        # two frames per sample on the GPU: the encoder writes its BEV map with a sample's frames side by side, which IS the
        # tensor the head would build with torch.cat (70 MB per step and the same again for the gradient's split)
        pair_bev = T == 2 and feats_all.is_cuda and os.environ.get("RSLO_PAIR_BEV", "1") != "0"
        # a head whose forward is replayed from a hipGraph reads a STATIC input map: the encoder writes its BEV map there
        bev_out = None
        if pair_bev:
            from rslo_amd import headgraph as _hg
            bev_out = _hg.static_input(self.odom_predictor, T, batch_size)
        bev, cov = self.middle_feature_extractor(feats_all, plan.indices, T * batch_size, plan=plan,
                                                 defer_cov=two_streams, bev_frames=T if pair_bev else 1,
                                                 **({} if bev_out is None else {"bev_out": bev_out}))
        if bev_out is not None and bev.shape != bev_out.shape:      # (another batch size than the graph's: a fresh map after all)
            bev_out = None
        cov_fn = cov if two_streams else None
        exchange = self.__dict__.get("_grad_exchange")      # data parallel: the head's gradient bucket leaves when the
        if exchange is not None:                            # gradient of the BEV map is complete (distributed_utils)
            exchange.watch(bev)
        if pair_bev:
            spatial_features = list(bev.split(bev.shape[1] // T, dim=1))         # channel-slice views, one per frame
            for f in spatial_features:
                f._pair_base = bev
        else:
            spatial_features = list(bev.split(batch_size, dim=0))
        if cov_fn is None:
            middle_conf_preds = list(cov.split([f.shape[0] for f in voxel_features], dim=0))
        self.end_timer("middle forward")

        if cov_fn is not None:
            # The covariance branch on its own stream.  It meets the rest of the network again only in the loss, and its
            # six level-1 / level-0 layers fill the GPU, while the head's half- and quarter-resolution stages (10 residual
            # blocks on 24 x 44 / 12 x 22 maps) launch about one workgroup per CU.  It is issued from INSIDE the head's
            # forward, after the encoder stages: the stream waits for the event recorded where those stages begin, so it
            # runs beside them in forward, and -- its graph nodes being created at that point -- autograd issues its
            # backward right before theirs (every node replays on the stream of its forward).
            cur = torch.cuda.current_stream(feats_all.device)
            side = _SIDE_STREAMS.get(feats_all.device)       # per device, outside the module: deepcopy / pickle of a
            if side is None:                                 # network must not meet a stream object
                from rslo_amd import streams as _streams      # (with several ranks: the stream the leaf work uses too)
                side = _SIDE_STREAMS[feats_all.device] = (_streams.side_stream(feats_all.device) if _streams.sharing()
                                                          else torch.cuda.Stream(feats_all.device))
            box = {}
            gate = torch.cuda.Event()

            def mark():
                gate.record(cur)

            def launch():
                side.wait_event(cov_fn.ready)
                side.wait_event(gate)
                cov_fn.source.record_stream(side)
                with torch.cuda.stream(side):
                    box["cov"] = cov_fn()
            if os.environ.get("RSLO_COV_STREAM", "1") != "2":      # "2": issued behind the whole head, no gate (A/B runs)
                self.odom_predictor.__dict__["_side_work"] = (mark, launch)
        # the head's training pass as two replayed hipGraphs (rslo_amd/headgraph.py) once its shape has been seen twice
        from rslo_amd import headgraph
        preds_dict = None
        if pair_bev and headgraph.wanted(self.odom_predictor, bev, T):
            side_work = self.odom_predictor.__dict__.pop("_side_work", None)
            if side_work is not None and _GRAPH_COV == "before":       # no hook points inside a graph: the branch starts beside
                side_work[0]()                                          # the head's first stage
                side_work[1]()
            preds_dict = headgraph.run(self.odom_predictor, bev, T)
            if preds_dict is None and side_work is not None and "cov" not in box:
                self.odom_predictor.__dict__["_side_work"] = side_work
        if preds_dict is None:
            preds_dict = self.odom_predictor(spatial_features, tq_map_gt=example.get("tq_maps", [None])[0])
        if cov_fn is not None:
            if "cov" not in box:        # a head without the hook points (registry variant): run the branch now
                self.odom_predictor.__dict__.pop("_side_work", None)
                launch()
            cov = box["cov"]
            from rslo_amd import streamprobe
            streamprobe.wait("cov_branch_forward", cur, lambda: cur.wait_stream(side))
            cov.record_stream(cur)
            middle_conf_preds = list(cov.split([f.shape[0] for f in voxel_features], dim=0))
        with torch.no_grad():
            sums = preds_dict.pop("_bev_sums", None)       # [B, T, H, W] per-frame channel sums the head already made
            if sums is not None and os.environ.get("RSLO_BEV_DISPLAY", "1") != "0":
                from rslo_amd import capi       # mask + both normalised maps in two launches (same bits as the lines below)
                preds_dict["feature_mask"], preds_dict["middle_feature"] = capi.bev_display(
                    sums, spatial_features[0].shape[1])
            else:
                if sums is not None:
                    preds_dict["feature_mask"] = (sums.sum(dim=1, keepdim=True) != 0).float()
                    disp = [sums[:, t:t + 1] / float(spatial_features[t].shape[1]) for t in range(T)]
                else:
                    preds_dict["feature_mask"] = (torch.cat(spatial_features, dim=1).sum(dim=1, keepdim=True) != 0).float()
                    disp = [f.mean(dim=1, keepdim=True) for f in spatial_features]
                preds_dict["middle_feature"] = [(d - d.min()) / (d.max() - d.min() + 1e-12) for d in disp]
        preds_dict["middle_conf_preds"] = middle_conf_preds
        preds_dict["voxel_features"] = voxel_features
        preds_dict["voxel_coords"] = coors
        preds_dict["normal_preds"] = []
        return preds_dict

    def forward(self, example):
        voxels, num_points, coors = example["voxels"], example["num_points"], example["coordinates"]
        if len(num_points[0].shape) == 2:   # padded multi-gpu layout: [B, maxN, ...] + num_voxels
            vb, nb, cb = [], [], []
            for t in range(len(voxels)):
                nv = example["num_voxels"][t].cpu().numpy().reshape(-1)
                vb.append(torch.cat([voxels[t][i, :n] for i, n in enumerate(nv)], 0))
                nb.append(torch.cat([num_points[t][i, :n] for i, n in enumerate(nv)], 0))
                cb.append(torch.cat([coors[t][i, :n] for i, n in enumerate(nv)], 0))
            voxels, num_points, coors = vb, nb, cb
        batch_size_dev = example["num_voxels"][0].shape[0]
        throttle = self.training and voxels[0].is_cuda and _HOST_LEAD > 0
        if throttle:
            # The issuing thread runs at most _HOST_LEAD forward passes ahead of the GPU (a sleeping wait on the event
            # recorded behind an earlier forward).  Nothing else bounds it -- the step has no host read -- and a thread
            # that fills the launch queue spins inside hipLaunchKernel and costs GPU time: 14.1 vs 13.7 ms per step
            # measured.  RSLO_HOST_LEAD=0: unbounded.
            ring = _LEAD_EVENTS.get(self)          # kept outside the module (events do not deep-copy), keyed weakly: no
            if ring is None:                       # stale ring under a recycled id() once a network is freed
                ring = _LEAD_EVENTS[self] = []
            if len(ring) >= _HOST_LEAD:
                t0, c0 = time.perf_counter(), time.thread_time()
                ring[-_HOST_LEAD].synchronize()                       # a sleeping wait (blocking-sync events)
                _LEAD_WAIT[0] += time.perf_counter() - t0
                _LEAD_WAIT[1] += time.thread_time() - c0
        if self.training and torch.is_grad_enabled() and voxels[0].is_cuda:
            from rslo.layers import hip_conv2d      # the head's weight operands: split beside the encoder's forward
            dev_ = voxels[0].device
            side_ = _SIDE_STREAMS.get(dev_)          # the covariance branch's stream (idle until the head starts)
            if side_ is None:
                from rslo_amd import streams as _streams
                side_ = _SIDE_STREAMS[dev_] = _streams.side_stream(dev_) if _streams.sharing() else torch.cuda.Stream(dev_)
            hip_conv2d.presplit_early(self.odom_predictor, dev_, side_)
        # one multi-tensor add for the num_batches_tracked buffers of every normalisation layer (ROCm apex stand-in only)
        with getattr(_apex_parallel, "defer_batch_counts", contextlib.nullcontext)():
            preds_dict = self.network_forward(voxels, num_points, coors, batch_size_dev, example=example)

        if self.training:
            if voxels[0].is_cuda:       # where the loss begins on this stream: the structure plan of a coming batch may be
                g = torch.cuda.Event()  # gated here (rslo_amd/workload.py RSLO_PLAN_GATE=loss)
                g.record(torch.cuda.current_stream(voxels[0].device))
                GATE_EVENTS[self] = g        # outside the module: events neither deep-copy nor pickle
            ret = self.loss(example, preds_dict)
            if throttle:
                ev = torch.cuda.Event(blocking=True)
                ev.record(torch.cuda.current_stream(voxels[0].device))
                ring.append(ev)
                del ring[:-(_HOST_LEAD + 1)]
            extras = {
                "middle_feature": preds_dict["middle_feature"], "feature_mask": preds_dict["feature_mask"],
                "t_conf": preds_dict.pop("t_conf", None), "r_conf": preds_dict.pop("r_conf", None),
                "pyramid_motion": preds_dict.pop("pyramid_motion", None),
                "dynamic_sigma": preds_dict.pop("dynamic_sigma", -1),
                "transformed_inputs": preds_dict.pop("transformed_inputs", None),
                "tq_map_g": preds_dict.pop("tq_map_g", None), "local_motion": preds_dict.pop("local_motion", None),
                "down_masks": preds_dict.pop("down_masks", None),
                "middle_conf_preds": list(preds_dict["middle_conf_preds"]),
            }
            ret.update(_cpu_tree(extras) if self.cpu_extras else _detach_tree(extras))
            return ret

        t_pred, r_pred = preds_dict["translation_preds"], preds_dict["rotation_preds"]
        if isinstance(t_pred, (list, tuple)):
            t_pred = t_pred[-1]
        if isinstance(r_pred, (list, tuple)):
            r_pred = r_pred[-1]
        out = {"translation_preds": t_pred.detach(), "rotation_preds": r_pred.detach()}
        if self.testing:
            out["middle_conf_preds"] = _detach_tree(list(preds_dict["middle_conf_preds"]))
            out["voxel_features"] = _detach_tree(preds_dict["voxel_features"])
            out["normal_preds"] = preds_dict["normal_preds"]
            out["tq_map_g"] = preds_dict["tq_map_g"].detach()
            out["pyramid_motion"] = _detach_tree(preds_dict["pyramid_motion"])
            out["t_conf"] = preds_dict["t_conf"].detach()
            out["r_conf"] = preds_dict["r_conf"].detach()
            out["normal_gt"] = example.get("normal_gt", None)
        return out

    # ------------------------------------------------------------------ loss
    def gen_tq_maps(self, odometries, spatial_size, pc_range, cubic_tq_map=False):
        if len(spatial_size) == 2:
            spatial_size = [1] + list(spatial_size)
        grid_size = np.array(list(spatial_size[::-1]))
        pc_range = np.asarray(pc_range)
        voxel_size = (pc_range[3:] - pc_range[0:3]) / grid_size
        ssize = grid_size if cubic_tq_map else grid_size[:2]
        origin_loc = ((0 - pc_range[0]) / (pc_range[3] - pc_range[0]) * grid_size[0],
                      (pc_range[4] - 0) / (pc_range[4] - pc_range[1]) * grid_size[1],
                      (0 - pc_range[2]) / (pc_range[5] - pc_range[2]) * grid_size[2])
        maps = [generate_pointwise_local_transformation_tch(tq, spatial_size=ssize, origin_loc=origin_loc,
                                                            voxel_size=voxel_size, inv_trans_factor=-1)
                for tq in odometries]
        return [torch.stack(maps, dim=0)]

    @amp.float_function
    def loss(self, example, preds_dict):
        T_preds, R_preds = preds_dict["translation_preds"], preds_dict["rotation_preds"]
        if not isinstance(T_preds, (list, tuple)):
            T_preds = [T_preds]
        if not isinstance(R_preds, (list, tuple)):
            R_preds = [R_preds]
        self.start_timer("create_loss forward")
        res = self.create_loss(
            preds_dict, example, self._translation_loss, self._rotation_loss,
            pyramid_rotation_loss=self._pyramid_rotation_loss,
            pyramid_translation_loss=self._pyramid_translation_loss, consistency_loss=self._consistency_loss,
            raw_tail=self.fused_loss_tail)
        if isinstance(res, dict):       # the partial losses: reductions, loss weights and the total in ONE launch
            n = res["pyr_loss_b"].shape[0] if res["pyr_loss_b"] is not None else 0
            total, terms = losses.loss_tail(res, [self._pyloss_exp_w_base ** (n - i) for i in range(n)])
            self.end_timer("create_loss forward")
            return {"loss": total, "translation_loss": terms[1:2], "rotation_loss": terms[2:3],
                    "pyramid_loss": terms[3:4], "C_loss": terms[4:5],
                    "translation_preds": T_preds[0].detach(), "rotation_preds": R_preds[0].detach()}
        t_loss, r_loss, py_T, py_R, C_loss = res
        n = len(py_T)
        if getattr(self, "_py_scaled", None) is not None:     # fused pyramid path: one weighted sum over [L,2]
            w = losses._const([self._pyloss_exp_w_base ** (n - i) for i in range(n)], T_preds[0].device)
            pyramid_loss = (w * self._py_scaled.sum(1)).sum().reshape(1)
            self._py_scaled = None
        else:
            pyramid_loss = torch.zeros([1], dtype=T_preds[0].dtype, device=T_preds[0].device)
            for i, (tl, rl) in enumerate(zip(py_T, py_R)):
                pyramid_loss = pyramid_loss + self._pyloss_exp_w_base ** (n - i) * (tl + rl)
        loss = t_loss + r_loss + pyramid_loss + C_loss
        self.end_timer("create_loss forward")
        return {"loss": loss, "translation_loss": t_loss.detach(), "rotation_loss": r_loss.detach(),
                "pyramid_loss": pyramid_loss.detach(), "C_loss": C_loss.detach(),
                "translation_preds": T_preds[0].detach(), "rotation_preds": R_preds[0].detach()}

    @amp.float_function
    def create_loss(self, preds_dict, example, translation_loss, rotation_loss, pyramid_translation_loss=None,
                    pyramid_rotation_loss=None, pyramid_preds=None, consistency_loss=None, raw_tail=False):
        """raw_tail=True (GPU, shipped loss configuration): returns the partial losses as a dict for
        losses.loss_tail instead of the five assembled terms."""
        translation_preds, rotation_preds = preds_dict["translation_preds"], preds_dict["rotation_preds"]
        if not isinstance(translation_preds, (list, tuple)):
            translation_preds = [translation_preds]
        if not isinstance(rotation_preds, (list, tuple)):
            rotation_preds = [rotation_preds]
        dtype, device = translation_preds[0].dtype, translation_preds[0].device
        pyramid_preds = preds_dict["pyramid_motion"]
        example["icp_odometry"] = example["icp_odometry"].view(-1, 7)
        translation_targets = example["icp_odometry"][:, :3]
        rotation_targets = example["icp_odometry"][:, 3:]
        step = self.get_global_step()

        if translation_loss._loss_weight == 0:
            self.warm_flag = True
        if self.warm_flag:
            warm_weight = 1.0 / (0.001 * step + 1) if step < 1500 else 0
            translation_loss._loss_weight = warm_weight
            rotation_loss._loss_weight = warm_weight
        else:
            warm_weight = 0

        res_r = res_t = None
        raw_pair = None
        AW = losses.AdaptiveWeightedL2Loss
        raw_tail = bool(
            raw_tail and device.type == "cuda" and dtype == torch.float32 and self.fused_pyramid
            and len(translation_preds) == 1 and len(rotation_preds) == 1 and translation_preds[0].shape[-1] == 3
            and rotation_preds[0].shape[-1] == 4 and self.odom_predictor._cubic_pred_height == 0
            and all(isinstance(m, AW) and m.focal_gamma == 0 for m in (translation_loss, rotation_loss))
            and all(m is None or (isinstance(m, AW) and m.focal_gamma == 0)
                    for m in (pyramid_translation_loss, pyramid_rotation_loss))
            and (pyramid_translation_loss is None) == (pyramid_rotation_loss is None)
            and (consistency_loss is None or getattr(consistency_loss, "focal_gamma", 0) == 0))
        C_loss = None if raw_tail else torch.zeros([1], dtype=dtype, device=device)      # the fused tail has its own
        if consistency_loss is not None:
            if len(preds_dict["middle_conf_preds"]) == 0:
                # the reference has no working behaviour here: it keeps point_confs = None (voxel_odom_net.py:628) and
                # indexes it in the consistency_loss call (:702-703) -> TypeError on the first step
                raise NotImplementedError("hier_points supervision without a covariance head (SURVEY.md 8f-4): the "
                                          "reference itself fails in this branch (point_confs is None)")
            feats = preds_dict["voxel_features"]
            B = example["num_voxels"][0].shape[0]
            # GPU, more than one sample: one launch per frame selects the columns, joins the covariance rows and pads
            one_launch = device.type == "cuda" and B > 1 and all(f.dtype == torch.float32 for f in feats)
            # xyz + normal columns (intensity dropped); slices, not an index list (no host->device index upload)
            if one_launch:
                pass
            elif feats[0].shape[1] > 6:
                feats = [torch.cat([f[:, 0:3], f[:, 4:7]], 1) for f in feats]
            else:
                feats = [f[:, 0:6] for f in feats]
            # rows of sample b inside frame t (frames hold the samples back to back)
            if B == 1:
                counts = [[f.shape[0]] for f in feats]
            else:
                counts = [[int(v) for v in example["num_voxels"][t].reshape(-1).tolist()] for t in range(len(feats))]
            T_ = len(feats)
            confs_all = preds_dict["middle_conf_preds"]
            # the reference truncates every frame of a sample to the shortest one (voxel_odom_net.py:646-651)
            lens = [min(counts[t][b] for t in range(T_)) for b in range(B)]
            if B == 1:
                points = [feats[t][:lens[0]][None] for t in range(T_)]
                confs = [confs_all[t][:lens[0]][None] for t in range(T_)]
                cnt_dev = cnt_host = None
            else:
                # all samples as ONE zero-padded batch [B, Lmax, .] per frame: a single row gather per frame
                # (row offsets / lengths go up as one small pinned upload -- no per-sample slicing and padding ops)
                Lmax = max(lens)
                meta = pair_rows_meta(counts)
                if device.type == "cuda":      # pinned + async: a pageable upload would drain the stream
                    meta = meta.pin_memory().to(device, non_blocking=True)
                points, confs, normals = [], [], []
                for t in range(T_):
                    if one_launch:
                        pt, nr, cf = losses.pair_rows(feats[t], confs_all[t], meta[t], meta[T_], Lmax)
                        normals.append(nr)
                    else:
                        both = losses.pad_rows(torch.cat([feats[t], confs_all[t]], 1), meta[t], meta[T_], Lmax)
                        nf = feats[t].shape[1]        # one split: its backward is one cat, not 2 x (zeros + copy) + add
                        pt, cf = both.split([nf, both.shape[-1] - nf], dim=-1)
                    points.append(pt)
                    confs.append(cf)
                npairs_ = T_ * (T_ - 1) // 2
                cnt_host = [n for n in lens for _ in range(npairs_)]
                cnt_dev = meta[T_]
                if npairs_ > 1:
                    cnt_dev = cnt_dev.repeat_interleave(npairs_)
            pts1, pts2 = create_cycle_constraint_data(points, 1)      # [B * npairs, L, 6], sample-major
            cov1, cov2 = create_cycle_constraint_data(confs, 1)
            if one_launch:      # xyz and normals arrived as separate contiguous tensors
                xyz1, xyz2 = pts1, pts2
                nrm1, nrm2 = create_cycle_constraint_data(normals, 1)
            else:
                xyz1, nrm1, xyz2, nrm2 = pts1[:, :, :3], pts1[:, :, 3:], pts2[:, :, :3], pts2[:, :, 3:]

            weights = [0.01, 0.01, 0.05, 0.1, 1]
            for R_pred, T_pred, weight in zip(rotation_preds, translation_preds, weights[-len(translation_preds):]):
                if R_pred.shape[-1] == 9:
                    R_pred = R_pred.reshape(-1, 3, 3)
                else:
                    R_pred = losses.quat_wxyz_to_rot(R_pred)
                if step <= 1500:   # warm-up: the consistency loss sees the identity pose
                    R_pred = torch.eye(3, device=device, dtype=dtype).expand(R_pred.shape[0], 3, 3).contiguous()
                    T_pred = torch.zeros_like(T_pred)
                icp_iter = self.icp_iter if step > 1500 else 5
                # the points/normals are network inputs: only the pose receives a gradient through the move
                p2_moved = losses.rigid_move(xyz2, R_pred, T_pred)
                n2_moved = losses.rigid_move(nrm2, R_pred.detach())
                lb, res_r, res_t = consistency_loss.pair_losses(
                    xyz1, p2_moved, cov_pred=cov1, cov_target=cov2, R_pred=R_pred, t_pred=T_pred,
                    normal_pred=nrm1.detach(), normal_target=n2_moved.detach(), icp_iter=icp_iter,
                    counts=cnt_dev, counts_host=cnt_host)
                if raw_tail:
                    raw_pair = (lb, (1 - warm_weight) * weight * consistency_loss._loss_weight)
                    continue
                l = consistency_loss._loss_weight * consistency_loss.reduce(lb)
                C_loss = C_loss + (1 - warm_weight) * weight * l

        tq_targets = None
        if res_r is not None and res_t is not None:
            if raw_tail:       # the (t*, q*) rows of the pyramid supervision come out of the same launch
                rotation_targets, translation_targets, tq_targets = losses.icp_pose_targets(res_r, res_t, R_pred, T_pred,
                                                                                            with_tq=True)
            else:
                rotation_targets, translation_targets = losses.icp_pose_targets(res_r, res_t, R_pred, T_pred)

        if raw_tail:
            raw = {"t_pred": translation_preds[0], "q_pred": rotation_preds[0], "t_tgt": translation_targets,
                   "q_tgt": rotation_targets, "pyr_loss_b": None,
                   "pair_loss": raw_pair[0] if raw_pair is not None else None,
                   "alphas": [translation_loss.alpha, rotation_loss.alpha,
                              (pyramid_translation_loss or translation_loss).alpha,
                              (pyramid_rotation_loss or rotation_loss).alpha,
                              consistency_loss.alpha if consistency_loss is not None else translation_loss.alpha],
                   "w": [translation_loss._loss_weight, rotation_loss._loss_weight,
                         pyramid_translation_loss._loss_weight if pyramid_translation_loss is not None else 0.0,
                         pyramid_rotation_loss._loss_weight if pyramid_rotation_loss is not None else 0.0,
                         raw_pair[1] if raw_pair is not None else 0.0]}
            levels = [(pp[0], pp[1]) if isinstance(pp, (tuple, list)) else (pp, None) for pp in pyramid_preds]
            if tq_targets is None:
                tq_targets = torch.cat([translation_targets, rotation_targets], dim=-1).reshape(-1, 7)
            example["tq_targets"] = tq_targets
            if pyramid_translation_loss is not None and len(levels) > 0:
                if not all(m is not None and p.dim() == 4 for p, m in levels):
                    raise NotImplementedError("pyramid levels without masks are outside the fused loss assembly")
                H0, W0 = (int(v) for v in levels[-1][0].shape[2:])
                _, vs, origin = _tq_map_geometry([1, H0, W0], self.odom_predictor.point_cloud_range)
                raw["pyr_loss_b"] = losses._PyramidL2Fn.apply(
                    tq_targets.detach().contiguous().float(), (H0, W0, origin, vs),
                    [m.detach().contiguous().float() for _, m in levels], *[p.float() for p, _ in levels])
            return raw
        if (len(translation_preds) == 1 and len(rotation_preds) == 1
                and isinstance(translation_loss, losses.AdaptiveWeightedL2Loss)
                and isinstance(rotation_loss, losses.AdaptiveWeightedL2Loss)
                and translation_preds[0].shape[-1] == 3 and rotation_preds[0].shape[-1] == 4):
            T_loss, R_loss = losses.pose_l2_losses(translation_preds[0], translation_targets, rotation_preds[0],
                                                   rotation_targets, translation_loss, rotation_loss)
        else:
            T_loss = sum(translation_loss(p, translation_targets) for p in translation_preds)
            R_loss = sum(rotation_loss(p, rotation_targets) for p in rotation_preds)
        if pyramid_translation_loss is None or pyramid_rotation_loss is None:
            return T_loss, R_loss

        self._py_scaled = None
        tq_targets = torch.cat([translation_targets, rotation_targets], dim=-1).reshape(-1, 7)
        example["tq_targets"] = tq_targets
        levels = [(pp[0], pp[1]) if isinstance(pp, (tuple, list)) else (pp, None) for pp in pyramid_preds]
        cubic = self.odom_predictor._cubic_pred_height > 0
        fused = (self.fused_pyramid and len(levels) > 0 and not cubic and tq_targets.is_cuda
                 and all(m is not None and p.dim() == 4 for p, m in levels))
        if fused:
            # targets are recomputed per cell inside the kernel: the [B,7,H,W] map of gen_tq_maps is never built
            H0, W0 = (int(v) for v in levels[-1][0].shape[2:])
            _, vs, origin = _tq_map_geometry([1, H0, W0], self.odom_predictor.point_cloud_range)
            scaled = losses.pyramid_l2_losses([p for p, _ in levels], [m for _, m in levels], tq_targets,
                                              (H0, W0, origin, vs), pyramid_translation_loss, pyramid_rotation_loss)
            self._py_scaled = scaled
            cols = [row.split(1) for row in scaled.unbind(0)]      # [L,2] -> per level ([1], [1])
            pyramid_T_losses = [c[0] for c in cols]
            pyramid_R_losses = [c[1] for c in cols]
            return T_loss, R_loss, pyramid_T_losses, pyramid_R_losses, C_loss

        if len(pyramid_preds) > 0:
            example["tq_maps"] = self.gen_tq_maps(
                tq_targets, spatial_size=pyramid_preds[-1][0].shape[2:],
                pc_range=self.odom_predictor.point_cloud_range, cubic_tq_map=cubic)
        pyramid_targets = list(example["tq_maps"])
        pyramid_T_losses, pyramid_R_losses = [], []
        for pred, pred_mask in levels:
            T_p, R_p = pred[:, :3], pred[:, 3:]
            T_tgt, R_tgt = pyramid_targets[0][:, :3], pyramid_targets[0][:, 3:]
            if T_tgt.shape != T_p.shape:
                T_tgt = F.interpolate(T_tgt, size=T_p[0, 0].shape, mode="nearest")
            if R_tgt.shape != R_p.shape:
                R_tgt = F.interpolate(R_tgt, size=T_p[0, 0].shape, mode="nearest")
            pyramid_T_losses.append(pyramid_translation_loss(T_p, T_tgt, mask=pred_mask[:, :1]))
            pyramid_R_losses.append(pyramid_rotation_loss(R_p, R_tgt, mask=pred_mask[:, -1:]))
        return T_loss, R_loss, pyramid_T_losses, pyramid_R_losses, C_loss
