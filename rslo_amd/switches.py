"""Every environment variable the PYTHON host of this package reads, in one place (the C library reads none: its
launch-code switches are `rslo_tuning_set`).  `tests/test_cabi.py::test_every_environment_switch_is_registered` fails when
a module reads an `RSLO_*` variable that is not listed here, so this table is the complete list.

Three kinds:
  location   where things are
  mode       documented modes of operation a user may want
  path       "=0" puts one fused / hand-written stage back on its unfused formulation (torch ops or the layer-by-layer
             nodes): the equivalence tests compare the two; not performance settings
"""

SWITCHES = {
    # name: (default, kind, meaning)
    "RSLO_HIPCC_EXTRA": ("", "location", "extra hipcc flags of rslo_amd/build.py (experiments: -DSPC_PREW_ALL=1); part of the source hash"),
    "RSLO_HIP_LIB": ("", "location", "path of librslo_hip.so (default: next to the package)"),
    "RSLO_REFERENCE_ROOT": ("", "location", "a reference checkout whose non-hot-path modules (protos, logging, ...) are forwarded to"),
    "RSLO_SYNCBN_EXCHANGE": ("auto", "mode", "SyncBN statistics exchange: auto | device | host | rccl (rslo_amd/peer.py)"),
    "RSLO_SYNCBN_FUSED_PEER": ("1", "mode", "0: multi-rank SyncBN as statistics kernel -> exchange kernel -> apply kernel on every map, instead of ONE kernel per direction that meets its peers per channel (rslo_bn2d_fwd_peer) on the register-cached maps"),
    "RSLO_FORCE_SYNCBN_PATH": ("0", "mode", "1: a one-rank process group runs the multi-rank SyncBN path (world-size-1 peer comm): what a rank of an N > 1 job executes, measurable on one GPU (bench.py multirank_path)"),
    "RSLO_SYNCBN_MAX_BATCH": ("0", "mode", "largest per-rank batch of a job whose ranks may hold different batch sizes: the single-launch / three-launch choice of every SyncBN layer is then made for that size on every rank (0: each rank's own batch, equal on all ranks)"),
    "RSLO_PEER_TIMEOUT_MS": ("600000", "mode", "how long a SyncBN exchange waits for a peer before it poisons the statistics with NaN and raises the comm's status (polled once per step)"),
    "RSLO_SYNCBN_HP_GROUP": ("0", "mode", "1: a dedicated high-priority RCCL group for the SyncBN collectives (only when they are collectives)"),
    "RSLO_OVERLAP_GRADS": ("1", "mode", "0: one gradient bucket after backward instead of the overlapped head bucket"),
    "RSLO_WGRAD_STREAM": ("1", "mode", "0: dense weight gradients on the issuing stream instead of the leaf stream (rslo_amd/streams.py)"),
    "RSLO_DEFER_WGRAD_REDUCE": ("8", "mode", "the partials -> gradient stage of the weight gradients: n >= 2: one launch per n layers and stream (rslo_amd/streams.py, csrc/wgrad_reduce.hip; same bits), 1: one launch per stream at the end of the backward pass, 0: one launch per layer"),
    "RSLO_SHARE_SIDE_STREAM": ("auto", "mode", "dense weight gradients on the covariance branch's stream (three streams of ours, the fourth slot left to the collective library's: a fifth active stream halves the step's speed, profiles/r06_fifth_stream.txt): auto = in a process with an initialised process group, 1 = always, 0 = never"),
    "RSLO_COV_STREAM": ("1", "mode", "0: covariance branch on the training stream; 2: issued behind the whole head (A/B)"),
    "RSLO_HOST_LEAD": ("1", "mode", "forward passes the issuing thread may run ahead of the GPU (0: unbounded)"),
    "RSLO_NATIVE_PLAN": ("1", "mode", "0: Python-issued voxelization + rulebook plan instead of rslo_plan_encoder"),
    "RSLO_PRESPLIT_EARLY": ("1", "mode", "0: head weight operands split in front of the head instead of beside the encoder"),
    "RSLO_PLAN_GATE": ("loss", "mode", "where a structure-plan job of a coming batch may start on the GPU: 'loss' where the current step's loss begins, 'head' at the head's small-map stages, 'fwd_end' behind the loss, 'none' at once"),
    "RSLO_HEAD_GRAPH": ("fwd", "mode", "0: the head issued launch by launch; fwd: the BEV head's training forward replayed from one hipGraph, its backward issued launch by launch over the capture's retained autograd graph; 1 / full: backward replayed as well (rslo_amd/headgraph.py; same bits as the eager pass)"),
    "RSLO_HEAD_GRAPH_INPUT": ("direct", "mode", "direct: the encoder's dense() writes the BEV map into the static input of the head's replayed graph; copy: into a fresh tensor that is then copied there (70 MB per step; A/B runs)"),
    "RSLO_HEAD_GRAPH_COV": ("before", "mode", "where the covariance branch's forward is issued when the head is a replayed graph: 'before' or 'after' the head"),
    "RSLO_INFER_PLAN_STREAMS": ("1", "mode", "side streams the inference runner (rslo_amd/inference.py) issues the coming scans' structure plans on, round-robin"),
    "RSLO_INFER_PLAN_PRIORITY": ("", "mode", "HIP stream priority of the inference runner's structure-plan stream(s) (default: the runtime's default; measured without effect on C2)"),
    "RSLO_PREFETCH_PRIORITY": ("", "mode", "HIP stream priority of the structure-plan stream (default: lowest)"),
    "RSLO_SWITCH_INTERVAL": ("0.0002", "mode", "interpreter switch interval while helper threads issue GPU work"),
    "RSLO_SPCONV_SPLIT": ("1", "path", "0: 32/64-channel sparse layers on the fp32-MFMA kernels instead of the split-bf16 ones"),
    "RSLO_ROW_ORDER": ("1", "path", "0: no mask-sorted tile order on the transposed tables"),
    "RSLO_CONV2D_PASSES": ("wfd", "path", "which passes of the dense 3x3 layers run on conv2d.hip (w, f, d); empty: the library"),
    "RSLO_CONV2D_S2": ("1", "path", "0: stride-2 layers on the library"),
    "RSLO_FUSED_BN": ("1", "path", "0: torch BatchNorm / SyncBatchNorm instead of bn2d.hip (auto: only with > 1 rank)"),
    "RSLO_FUSED_BLOCK": ("1", "path", "0: a BasicBlock as layer-by-layer autograd nodes instead of one node"),
    "RSLO_FUSED_HEAD_TAIL": ("1", "path", "0: the head's element-wise tail as torch ops instead of headtail.hip"),
    "RSLO_FUSED_LOSS_TAIL": ("1", "path", "0: loss assembly as torch ops instead of k_loss_tail_*"),
    "RSLO_FUSED_OPTIM": ("1", "path", "0: torch.optim.Adam without fused=True under the wrapper"),
    "RSLO_HIP_OPTIM": ("1", "path", "0: torch clip_grad_norm_ + Adam instead of optim.hip"),
    "RSLO_CAT_UPSAMPLE": ("1", "path", "0: torch cat + Upsample in front of a deblock"),
    "RSLO_PAIR_BEV": ("1", "path", "0: per-frame BEV maps + torch.cat instead of dense() writing the pair layout"),
    "RSLO_BEV_DISPLAY": ("1", "path", "0: logged display maps as torch ops"),
    "RSLO_HEAD_NHWC": ("0", "path", "bench.py only: the dense head in channels-last on the library path (measured slower)"),
}
