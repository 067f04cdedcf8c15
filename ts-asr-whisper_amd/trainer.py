"""Training-step driver for the DiCoW hot path: flat parameter/gradient store, fused AdamW + clipping, data parallel.

The repository's counterpart of the reference harness (SURVEY.md section 8b last row): batch dict -> ``model(**batch)`` ->
``loss.backward()`` -> global-norm clip 1.0 -> AdamW with the two parameter groups of reference
src/models/containers.py:100-114 (base lr; ``prefixes_to_preheat`` parameters lr x fddt_lr_multiplier, weight decay 0)
-> cosine/linear-warmup schedule (configs/train/dicow_v3.yaml:66-68), keyword freezing (containers.py:80-97,
dicow_v3.yaml:6-7: every parameter whose name contains "decoder" is frozen).

Data parallel (reference: torchrun + DDP/NCCL, scripts/submit_slurm.sh:34): one process per GPU, full replica, disjoint
minibatch; trainable gradients live in ONE flat fp32 buffer laid out in backward-completion order, so that each encoder
layer's slice is all-reduced (RCCL over xGMI, ``torch.distributed`` backend "nccl") on a side HIP stream as soon as that
layer's backward has been enqueued, overlapped with the remaining backward.
"""
import collections
import logging
import math
import os
import time

import torch
import torch.distributed as dist

from . import ops
from . import tracing
from .engine import GradSink

F32 = torch.float32
log = logging.getLogger(__name__)


def freeze_by_keyword(model, frozen_keywords=("decoder",)):
    """reference containers.py:80-90: requires_grad = not any(keyword in name)."""
    for n, p in model.named_parameters():
        p.requires_grad_(not any(k in n for k in frozen_keywords))
    model.tie_weights()


def _backward_order(model):
    """Trainable encoder parameters grouped in the order their gradients complete during backward."""
    enc = model.model.encoder
    groups = []
    ctc_ids = set()
    if getattr(enc, "ctc_weight", 0.0) > 0.0:
        ctc = enc.ctc_parameters()
        ctc_ids = {id(p) for p in ctc}
        groups.append(("ctc", ctc))                        # the CTC branch's backward runs before the encoder's
    groups.append(("final_ln", list(enc.layer_norm.parameters())))
    nl = len(enc.layers)
    for i in range(nl - 1, -1, -1):
        ps = list(enc.layers[i].parameters())
        ps = [p for p in ps if p.dim() == 2] + [p for p in ps if p.dim() != 2]     # weight matrices first (FlatStore.zero_grad)
        if hasattr(enc, "fddts") and i < len(enc.fddts):
            ps += list(enc.fddts[i].parameters())
        if hasattr(enc, "ca_enrolls") and i < len(enc.ca_enrolls):
            ps += list(enc.ca_enrolls[i].parameters())
        groups.append((f"layer{i}", ps))
    tail = []
    if hasattr(enc, "initial_fddt"):
        tail += list(enc.initial_fddt.parameters())
    tail += [enc.embed_positions.weight] + list(enc.conv2.parameters()) + list(enc.conv1.parameters())
    groups.append(("stem", tail))
    return groups


# reference configs/base.yaml:18 and configs/train/se_dicow.yaml:13 (model.prefixes_to_preheat): the parameters that train in
# the preheat phase and get lr x fddt_lr_multiplier with weight decay 0 afterwards (containers.py:100-114) -- the FDDT modules,
# the CTC branch and the SE-DiCoW enrollment cross-attention.  Prefixes a model does not have match nothing.
REFERENCE_PREHEAT_PREFIXES = ("model.encoder.additional_layer", "model.encoder.additional_self_attention_layer", "model.encoder.lm_head",
                              "model.encoder.subsample_conv1", "model.encoder.subsample_conv2", "model.encoder.fddts",
                              "model.encoder.initial_fddt", "model.encoder.ca_enrolls")


class FlatStore:
    """Flat fp32 parameter / gradient / Adam-moment buffers for the trainable parameters.

    ``p.data`` and ``p.grad`` become views into the flat buffers; the engine accumulates gradients directly into
    ``p.grad`` (``p._direct_grad``), so autograd performs no extra accumulation pass and the fused optimizer and the
    bucketed all-reduce work on contiguous memory."""

    def __init__(self, model, preheat_prefixes=None):
        preheat_prefixes = REFERENCE_PREHEAT_PREFIXES if preheat_prefixes is None else tuple(preheat_prefixes)
        names = {id(p): n for n, p in model.named_parameters()}
        dev = next(model.parameters()).device
        self.segments = []            # (name, start, end) per backward group
        self.runs = []                # (start, end, is_preheat) contiguous optimizer runs
        entries, off, seen = [], 0, set()
        grouped = _backward_order(model)
        enc_ids = {id(p) for _, ps in grouped for p in ps}
        rest = [p for p in model.parameters() if p.requires_grad and id(p) not in enc_ids]
        if rest:
            grouped = [("decoder", rest)] + grouped      # decoder grads complete first (its backward runs first)
        for gname, ps in grouped:
            start = off
            for p in ps:
                if not p.requires_grad or id(p) in seen:
                    continue
                seen.add(id(p))
                n = p.numel()
                pre = any(names[id(p)].startswith(pp) for pp in preheat_prefixes)
                entries.append((p, off, n, pre))
                off += (n + 63) // 64 * 64
            if off > start:
                self.segments.append((gname, start, off))
        self.numel = off
        self.params = torch.zeros(off, dtype=F32, device=dev)
        self.grads = torch.zeros(off, dtype=F32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=F32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=F32, device=dev)
        self.entries = entries
        self._names = names
        for p, o, n, pre in entries:
            self.params[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
            p._direct_grad = p.grad
            pad_end = o + (n + 63) // 64 * 64
            if self.runs and self.runs[-1][2] == pre and self.runs[-1][1] == o:
                self.runs[-1] = (self.runs[-1][0], pad_end, pre)
            else:
                self.runs.append((o, pad_end, pre))
        self.n_trainable = sum(n for _, _, n, _ in entries)
        # Gradients the backward pass WRITES instead of accumulating into zeros: the encoder layers' weight matrices (628 M of
        # the 637 M trainable values of large-v3-turbo), each produced by exactly one weight-gradient GEMM per micro-batch
        # (engine.EncoderEngine.backward -> ops.TnGroup).  zero_grad(first_writer=True) skips their 2.5 GB fill and flags them;
        # the engine then runs the first micro-batch's GEMM with accumulate = 0 (no read of the old value either) and clears
        # the flag, so later micro-batches accumulate.  Everything else (vectors, FDDT, stem, decoder) is zeroed as before.
        enc = model.model.encoder
        over = {id(p) for lyr in enc.layers for p in lyr.parameters() if p.dim() == 2}
        self._over = [(p, o, n) for p, o, n, _ in entries if id(p) in over]
        keep, cur = [], 0
        for p, o, n in sorted(self._over, key=lambda t: t[1]):
            if o > cur:
                keep.append((cur, o))
            cur = max(cur, o + n)
        if cur < self.numel:
            keep.append((cur, self.numel))
        self._zero_ranges = keep                     # the complement of the overwritable matrices: ~one range per layer
        self._zero_views = None

    def fingerprint(self):
        """(parameter name, offset, numel) of every entry, in flat-buffer order: what an optimizer checkpoint must agree with."""
        return [(self._names[id(p)], int(o), int(n)) for p, o, n, _ in self.entries]

    def zero_grad(self, first_writer=False):
        """first_writer: only valid when every flagged matrix is certain to receive its gradient in the coming backward pass
        (the full training phase; TrainStep decides)."""
        if not first_writer or not self._over:
            self.grads.zero_()
            for p, _, _ in self._over:
                p._grad_overwrite = False
            return
        if self._zero_views is None:                  # the ~33 small ranges between the matrices: one multi-tensor fill, not 33 launches
            self._zero_views = [self.grads[a:b] for a, b in self._zero_ranges if b > a]
        if self._zero_views:
            torch._foreach_zero_(self._zero_views)
        for p, o, n in self._over:
            p._grad_overwrite = bool(p.requires_grad)
            if not p.requires_grad:                   # frozen after the store was built: no GEMM will write it, keep it zero
                self.grads[o:o + n].zero_()

    def settle_range(self, a, b):
        """The still-flagged matrices inside grads[a:b] (a bucket about to leave): zeroed and un-flagged ON THE COMPUTE STREAM, before the
        bucket's event is recorded -- zeros go on the wire, not last step's values, and the result does not depend on every rank
        having skipped the same matrices."""
        n_left = 0
        for p, o, n in self._over:
            if a <= o < b and getattr(p, "_grad_overwrite", False):
                self.grads[o:o + n].zero_()
                p._grad_overwrite = False
                n_left += 1
        return n_left

    def settle_first_writers(self):
        """After a backward pass under zero_grad(first_writer=True): a flagged matrix whose weight-gradient GEMM did NOT run
        (its flag is still set -- the encoder backward was not reached, or skipped that matrix) holds the PREVIOUS step's
        gradient; zero it so that neither the clip norm nor AdamW sees stale values.  Returns how many were settled."""
        n_left = 0
        for p, o, n in self._over:
            if getattr(p, "_grad_overwrite", False):
                self.grads[o:o + n].zero_()
                p._grad_overwrite = False
                n_left += 1
        return n_left


class FusedAdamW:
    """Two-group AdamW + global-norm clipping in fused HIP kernels over the FlatStore (no host synchronisation).

    Mirrors what the reference's harness does with ``torch.optim.AdamW`` (containers.py:100-114) under the HF Trainer:
      * the k-th optimizer step uses the scheduler's lambda(k - 1) (LambdaLR: the scheduler steps AFTER the optimizer), with
        ``get_cosine_schedule_with_warmup`` as the lambda (dicow_v3.yaml:66-67);
      * bias correction counts the updates each parameter has RECEIVED (torch keeps ``state['step']`` per parameter and
        skips parameters without a gradient), which matters for the staged unfreezing of trainers.py:122-137: a run
        that was frozen during the preheat phase starts at step 1 when it is unfrozen;
      * the preheat group has weight decay 0 whatever the base group uses (containers.py:109-111);
      * clipping: g *= min(1, max_norm / (||g||_2 + 1e-6)) over every gradient of the step."""

    def __init__(self, store, lr=2e-6, fddt_lr_multiplier=100.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=1.0, warmup_steps=0, max_steps=0, schedule="cosine"):
        self.s, self.lr, self.mult, self.betas, self.eps, self.wd = store, lr, fddt_lr_multiplier, betas, eps, weight_decay
        self.max_norm, self.warmup, self.max_steps, self.schedule = max_grad_norm, warmup_steps, max_steps, schedule
        self.t = 0                                        # optimizer steps taken (= HF state.global_step)
        self.run_t = [0] * len(store.runs)                # updates received per run
        self.gnorm_sq = torch.zeros(1, dtype=F32, device=store.params.device)
        # step counters and this step's scalars live on the DEVICE (dicow_adamw_hyper): counters[0] = optimizer steps,
        # counters[1 + i] = updates of run i, hyper[i] = {lr, 1 - beta1^t, 1 - beta2^t}; the host mirrors (t, run_t) advance
        # in lock-step and serve checkpointing and the phase switch
        dev = store.params.device
        self.counters = torch.zeros(1 + len(store.runs), dtype=torch.int32, device=dev)
        self.hyper = torch.zeros(len(store.runs), 3, dtype=F32, device=dev)
        self.is_pre = torch.tensor([int(pre) for _, _, pre in store.runs], dtype=torch.int32, device=dev)

    def sync_counters(self):
        """Host mirrors -> device counters (after load_state_dict)."""
        self.counters.copy_(torch.tensor([self.t] + list(self.run_t), dtype=torch.int32))

    def lr_at(self, k):
        """Learning rate of the k-th optimizer step (k = 1, 2, ...) -- what dicow_adamw_hyper computes on the device."""
        sched_step = k - 1
        if sched_step < self.warmup:
            return self.lr * sched_step / max(1, self.warmup)
        if self.schedule == "cosine" and self.max_steps > 0:
            prog = (sched_step - self.warmup) / max(1, self.max_steps - self.warmup)
            return self.lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        return self.lr

    def applied_lr(self):
        """Learning rates the LAST optimizer step applied, one per run, read back from the device (hyper[:, 0]: the scheduled
        rate x the group multiplier; what the reference's trainer logs per parameter group).  Synchronises: logging only."""
        return self.hyper[:, 0].detach().cpu().tolist()

    def advance(self, preheat_only=False):
        """Host mirrors of the device counters, one optimizer step on (the device side is launch())."""
        self.t += 1
        for i, (a, b, pre) in enumerate(self.s.runs):
            if not (preheat_only and not pre):
                self.run_t[i] += 1

    def launch(self, preheat_only=False):
        """Device side of one optimizer step (what a captured graph holds): counters + schedule, gradient norm, one fused
        clip + AdamW launch per run.  No host value that changes from step to step is a kernel argument."""
        s = self.s
        L_ = ops.L
        L_.call("dicow_adamw_hyper", self.counters.data_ptr(), self.hyper.data_ptr(), self.is_pre.data_ptr(), len(s.runs),
                int(preheat_only), float(self.lr), float(self.mult), int(self.warmup), int(self.max_steps),
                int(self.schedule == "cosine"), float(self.betas[0]), float(self.betas[1]), L_.stream())
        self.gnorm_sq.zero_()
        ops.sumsq(s.grads, self.gnorm_sq)                 # frozen runs hold zeros
        for i, (a, b, pre) in enumerate(s.runs):
            if preheat_only and not pre:
                continue
            ops.adamw_dev(s.params[a:b], s.grads[a:b], s.exp_avg[a:b], s.exp_avg_sq[a:b], self.hyper[i], self.betas[0], self.betas[1],
                          self.eps, 0.0 if pre else self.wd, gnorm_sq=self.gnorm_sq, max_norm=self.max_norm)

    def step(self, preheat_only=False):
        """preheat_only: update only the runs of the preheat group (the others are frozen: no gradient, no update)."""
        self.advance(preheat_only)
        self.launch(preheat_only)


# ------------------------------------------------------------------------------------------------ replica consistency
def _bit_checksum(t):
    """Order-independent 64-bit checksum of a tensor's BITS (any dtype of 1, 2, 4 or 8 bytes per element), as a Python int:
    two replicas that differ in any single bit differ in it.  Chunked: the int64 temporary stays small."""
    flat = t.detach().reshape(-1)
    if flat.numel() == 0:
        return 0
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[flat.element_size()]
    bits = flat.view(view)
    total, step = 0, 1 << 24
    for a in range(0, bits.numel(), step):
        c = bits[a:a + step].to(torch.int64)
        idx = torch.arange(a, a + c.numel(), device=c.device, dtype=torch.int64)
        total = (total + int((c * (2 * (idx % 1000003) + 1)).sum().item())) & 0xFFFFFFFFFFFFFFFF
    return total


_SYNC_COALESCE_BELOW = 64 << 20        # bytes: sync_replicas broadcasts larger tensors in place


def sync_replicas(model, store=None, process_group=None, mode="broadcast", src=0, extra=()):
    """Make (or check that) every rank of the data-parallel group holds rank ``src``'s model state -- what
    ``torch.nn.parallel.DistributedDataParallel`` does in its constructor (it broadcasts rank 0's parameters and buffers;
    the reference gets it from the HF Trainer's DDP wrap, scripts/submit_slurm.sh:34, configs/base.yaml:73).  Without it
    ranks that built the model from different RNG states, or loaded different files, train silently diverging replicas.

    mode "broadcast": rank ``src``'s flat parameter store (ONE collective for all trainable parameters), then the parameters
    outside the store (frozen ones) and the buffers coalesced per dtype, then ``extra`` tensors (Adam moments on resume).
    mode "verify": nothing is overwritten; a 64-bit checksum of every tensor's bits is all-gathered and a mismatch raises
    ``RuntimeError`` naming the ranks.  mode "none": no-op.  Returns the number of bytes put on the wire (0 for world 1 /
    "none"; the checksum exchange of "verify" counts its 8 bytes per rank).  Logged at INFO level."""
    if mode not in ("broadcast", "verify", "none"):
        raise ValueError(f"replica_sync must be 'broadcast', 'verify' or 'none', not {mode!r}")
    if mode == "none" or not dist.is_initialized():
        return 0
    world = dist.get_world_size(process_group)
    in_store = set()
    tensors = []
    if store is not None:
        in_store = {id(p) for p, _, _, _ in store.entries}
        tensors.append(store.params)
    seen = set()
    rest = []
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) in in_store or id(t) in seen or t.data_ptr() in seen:
            continue                                  # (tied weights share storage: once)
        seen.add(id(t))
        seen.add(t.data_ptr())
        rest.append(t.data)
    rest += [t for t in extra if t is not None]
    src_global = dist.get_global_rank(process_group, src) if process_group is not None else src
    if mode == "verify":
        mine = 0
        for t in tensors + rest:
            mine = (mine * 1000003 + _bit_checksum(t)) & 0x7FFFFFFFFFFFFFFF
        dev = tensors[0].device if tensors else (rest[0].device if rest else torch.device("cpu"))
        got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(got, torch.tensor([mine], dtype=torch.int64, device=dev), group=process_group)
        sums = [int(g.item()) for g in got]
        if len(set(sums)) != 1:
            bad = [r for r, v in enumerate(sums) if v != sums[src]]
            raise RuntimeError(f"data-parallel replicas differ at start-up: ranks {bad} do not hold rank {src}'s parameters / buffers "
                               f"(checksums {[hex(v) for v in sums]}); construct TrainStep with replica_sync='broadcast' or load the "
                               f"same checkpoint on every rank")
        log.info("replica check: %d ranks hold identical state (checksum %s)", world, hex(sums[0]))
        return 8 * world
    nbytes = 0
    if world > 1 or os.environ.get("DICOW_FORCE_REDUCE") == "1":
        for t in tensors:
            dist.broadcast(t, src=src_global, group=process_group)
            nbytes += t.numel() * t.element_size()
        # tensors above the threshold go out in place (the frozen decoder of large-v3 is 3.6 GB of fp32, Adam moments on resume
        # 5 GB: a torch.cat of those is a full-size temporary on top of model + optimizer state); only the small ones -- biases,
        # LayerNorm vectors, counters: thousands of launches otherwise -- are coalesced per dtype
        by_dtype = collections.OrderedDict()
        for t in rest:
            if t.numel() * t.element_size() >= _SYNC_COALESCE_BELOW and t.is_contiguous():
                dist.broadcast(t, src=src_global, group=process_group)
                nbytes += t.numel() * t.element_size()
            else:
                by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for (dt, dev), ts in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in ts]) if len(ts) > 1 else ts[0].reshape(-1).clone()
            dist.broadcast(flat, src=src_global, group=process_group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
            nbytes += flat.numel() * flat.element_size()
        log.info("replica sync: rank %d's state broadcast to %d ranks (%d bytes: %d in the flat store)", src, world, nbytes,
                 sum(t.numel() * t.element_size() for t in tensors))
    return nbytes


def nccl_pg_options():
    """Options for ``dist.init_process_group("nccl", pg_options=...)``: RCCL's own stream at high priority, so that it does not land on
    the hardware queue of the compute stream (see GradReducer.__init__).  None when the backend class is not available."""
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:
        return None


class GradReducer:
    """Bucketed gradient all-reduce on a side stream, one bucket per backward segment of the FlatStore."""

    def __init__(self, store, process_group=None):
        self.s = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # DICOW_FORCE_REDUCE=1 exercises the bucketed side-stream all-reduce even with a single rank (smoke test of
        # the RCCL code path on a 1-GPU box)
        self.force = os.environ.get("DICOW_FORCE_REDUCE") == "1" and dist.is_initialized()
        # HIGH-PRIORITY side stream: ROCm maps HIP streams onto a few hardware queues, and a normal-priority pool stream can share ITS
        # queue with the compute stream -- kernels of one queue never overlap, and the exchange then lengthens the step by exactly
        # its own busy time (round 6: rocprofv3 showed the bucket kernels and the GEMMs on the same Queue_Id, 0 % overlap,
        # profiles/r06_dp_emulated.txt).  Priority streams get queues of their own.  The process group should be created with
        # ProcessGroupNCCL.Options(is_high_priority_stream=True) for the same reason (bench.py does; see nccl_pg_options()).
        self.stream = None
        if (self.world > 1 or self.force) and torch.cuda.is_available():
            self.stream = torch.cuda.Stream(priority=-1 if os.environ.get("DICOW_REDUCER_STREAM_PRIORITY", "high") == "high" else 0)
        if self.stream is not None and self.world > 1:
            # RCCL runs one workgroup per channel beside the backward pass; the persistent GEMM's workgroups own a whole CU
            # each, so leave the channels their CUs (the all-reduce of 2.5 GB has the whole backward pass to hide in and
            # needs few channels: DICOW_RCCL_CUS, default 16; launchers should cap NCCL_MAX_NCHANNELS accordingly).
            from . import ops
            reserve = int(os.environ.get("DICOW_RCCL_CUS", "16"))
            ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            ops.set_gemm_cus(max(ncu - reserve, ncu // 2))
        # the mean of the ranks' gradients: ncclAvg where the backend has it (RCCL), sum + one division pass otherwise (gloo)
        self._avg_in_collective = bool(dist.is_initialized() and dist.get_backend(process_group) == "nccl")
        self.seg = {name: (a, b) for name, a, b in store.segments}
        self.pending = []
        self.time_exposed = False     # bench: record how long the compute stream waits for the side-stream buckets
        self._exposed = []
        # bench / SCALE runs: per bucket, when it became ready on the compute stream, when its all-reduce started on the side stream
        # (the difference = how long the bucket queued behind the ones before it) and how long the collective took
        self.time_buckets = False
        self._bucket_ev = []          # (name, ready, start, end) HIP events
        self._bucket_cpu = []         # (name, seconds) on the synchronous (gloo / CPU) path
        # one-GPU rehearsal: behind each bucket's (one-rank) all-reduce, the GPU-side load of an 8-rank one at this algorithm bandwidth
        g = os.environ.get("DICOW_EMULATE_FABRIC_GBPS")
        self.emulate_gbps = float(g) if g else 0.0
        self.emulate_workgroups = int(os.environ.get("DICOW_EMULATE_FABRIC_WGS", "16"))
        self.preheat_only = False     # staged freezing, phase 1: only the preheat runs carry gradients
        self.hold = False             # gradient accumulation: not the last micro-batch yet, nothing to exchange

    def segment_ready(self, name):
        """Called right after the segment's backward kernels have been enqueued on the current stream."""
        if (self.world == 1 and not self.force) or name not in self.seg or self.preheat_only or self.hold:
            return
        a, b = self.seg[name]
        self.s.settle_range(a, b)                     # a matrix of this bucket no weight-gradient GEMM wrote: zero it before it is sent
        if self.stream is None:                       # CPU / gloo tests: synchronous
            buf = self.s.grads[a:b]
            t0 = time.perf_counter()
            dist.all_reduce(buf, group=self.pg)
            buf.div_(self.world)
            if self.time_buckets:
                self._bucket_cpu.append((name, time.perf_counter() - t0))
            return
        ev = torch.cuda.Event(enable_timing=self.time_buckets)
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if self.time_buckets:
                e_start = torch.cuda.Event(enable_timing=True)
                e_start.record()
            buf = self.s.grads[a:b]
            if self._avg_in_collective:               # RCCL averages inside the collective: no second pass over the 2.5 GB
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(buf, group=self.pg)
                buf.div_(self.world)
            if self.emulate_gbps > 0.0 and buf.is_cuda:
                n = (b - a) // 4 * 4                  # (16-byte multiples; a tail of < 4 floats does not change the load)
                if n > 0:
                    ops.fabric_emulate(buf[:n], self.emulate_gbps, self.emulate_workgroups)
            if self.time_buckets:
                e_end = torch.cuda.Event(enable_timing=True)
                e_end.record()
                self._bucket_ev.append((name, ev, e_start, e_end))

    def _reduce_preheat_runs(self):
        """Phase 1 of the staged freezing: the frozen runs hold zeros, so only the (small) preheat runs are exchanged,
        after the backward pass, as one coalesced buffer."""
        runs = [(a, b) for a, b, pre in self.s.runs if pre]
        if not runs:
            return
        flat = torch.cat([self.s.grads[a:b] for a, b in runs])
        dist.all_reduce(flat, group=self.pg)
        flat.div_(self.world)
        off = 0
        for a, b in runs:
            self.s.grads[a:b].copy_(flat[off:off + b - a])
            off += b - a

    def finish(self):
        if self.preheat_only and (self.world > 1 or self.force):
            self._reduce_preheat_runs()
        if self.stream is not None:
            if self.time_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                torch.cuda.current_stream().wait_stream(self.stream)
                e1.record()
                self._exposed.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self.stream)

    def bucket_report(self, steps=1):
        """{buckets_per_step, start_lag_ms: {mean, max, sum_per_step}, busy_ms_per_step, slowest: [...]} from the time_buckets records:
        start lag = bucket ready on the compute stream -> its all-reduce starts on the side stream (it queued behind earlier buckets);
        busy = the collectives' own time.  On the synchronous (gloo / CPU) path only the collectives' wall time exists (lag 0)."""
        rows = []
        if self._bucket_ev:
            torch.cuda.synchronize()
            rows = [(n, r.elapsed_time(s_), s_.elapsed_time(e)) for n, r, s_, e in self._bucket_ev]
        elif self._bucket_cpu:
            rows = [(n, 0.0, dt * 1e3) for n, dt in self._bucket_cpu]
        self._bucket_ev, self._bucket_cpu = [], []
        if not rows:
            return None
        steps = max(1, steps)
        per = {}
        for n, lag, dur in rows:
            t = per.setdefault(n, [0, 0.0, 0.0])
            t[0] += 1; t[1] += lag; t[2] += dur
        worst = sorted(per.items(), key=lambda kv: -kv[1][1] / kv[1][0])[:3]
        return {"buckets_per_step": len(rows) // steps,
                "start_lag_ms": {"mean": round(sum(r[1] for r in rows) / len(rows), 4), "max": round(max(r[1] for r in rows), 4),
                                 "sum_per_step": round(sum(r[1] for r in rows) / steps, 3)},
                "busy_ms_per_step": round(sum(r[2] for r in rows) / steps, 3),
                "longest_queued": [{"bucket": n, "lag_ms": round(v[1] / v[0], 4), "collective_ms": round(v[2] / v[0], 4)} for n, v in worst]}

    def exposed_ms(self):
        """Mean per-step time the compute stream spent waiting for the gradient exchange (time_exposed = True steps)."""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        self._exposed = []
        return ms


_SPLIT_NOWAIT = os.environ.get("DICOW_SPLIT_NOWAIT") == "1"       # timing experiments only: the halves' gradient sums race
_SPLIT_LEAD_FIRST = os.environ.get("DICOW_SPLIT_LEAD_FIRST", "1") == "1"
_SPLIT_GROUP = int(os.environ.get("DICOW_SPLIT_GROUP", "2"))     # encoder layers per cross-stream wait


class SplitSync:
    """Order of the parameter-gradient writes when ONE batch runs as two half batches on two HIP streams (TrainStep(split_streams=True)).

    Both halves add into the same gradient buffers.  The LEADER half's backward is enqueued first (host order), it consumes the
    first-writer flags and records an event behind every segment of its backward pass (final LayerNorm, each encoder layer with its
    pooled weight-gradient launch, stem, decoder); the FOLLOWER half waits for the segment's event before it enqueues the same segment,
    so every gradient is `leader's share, then + follower's share` -- a fixed order, bit-reproducible run to run -- while the two
    streams otherwise run side by side, one encoder layer apart.  engine.EncoderEngine.backward / modeling._DecoderLossFn call
    ``enter(name)`` / ``done(name)``; outside a split step ``role`` is None and both are no-ops."""

    def __init__(self, group=_SPLIT_GROUP):
        self.role, self.ev, self.group, self.free_until = None, {}, max(1, int(group)), None

    def reset(self):
        self.role, self.ev, self.free_until = None, {}, None

    def _gate(self, name):
        """Encoder layers are gated in groups of ``group``: one cross-stream wait per group (a wait costs the follower ~40 us of
        queue latency whether or not it has to block).  Backward walks the layers downwards, so "the leader is through layer j"
        covers every layer above j: entering layer i the follower waits for the group's LOWEST layer and then runs down to it."""
        if not name.startswith("layer") or self.group == 1:
            return name
        i = int(name[5:])
        return f"layer{i - i % self.group}"

    def enter(self, name):
        if self.role == "follow" and not _SPLIT_NOWAIT:
            gate = self._gate(name)
            if gate == self.free_until:                 # already waited for this group's lowest layer
                return
            ev = self.ev.get(gate)
            if ev is None:
                raise RuntimeError(f"SplitSync: the leader half never finished segment {gate!r} -- the two halves' backward passes differ")
            torch.cuda.current_stream().wait_event(ev)
            self.free_until = gate

    def done(self, name):
        if self.role == "lead" and self._gate(name) == name:
            ev = torch.cuda.Event()
            ev.record()
            self.ev[name] = ev


class TrainStep:
    """model + FlatStore + FusedAdamW (+ GradReducer): ``loss = step(batch)``."""

    def __init__(self, model, lr=2e-6, fddt_lr_multiplier=100.0, weight_decay=0.0, max_grad_norm=1.0, warmup_steps=0,
                 max_steps=0, frozen_keywords=("decoder",), preheat_prefixes=None,
                 process_group=None, augmenter=None, use_fddt_only_n_steps=0, graph=False, use_fddt_only_n_epochs=0,
                 steps_per_epoch=None, replica_sync="broadcast", split_streams=False):
        self.model = model
        # split_streams=True: a batch of even size B >= 2 runs as two half batches on two HIP streams (same loss and gradients: the
        # halves' losses are weighted by their label counts, their gradients add in a fixed order -- SplitSync).  The HBM-bound row
        # kernels, the attention tails and the partial last rounds of one half fill under the other half's matrix kernels
        # (profiles/r06_streams.txt).  Sums of two M/2-row products are not bit-equal to one M-row product: opt-in.
        self.split_streams = bool(split_streams)
        self._split_sync = SplitSync()
        self._split_side = None
        # preheat_prefixes=None: the reference's model.prefixes_to_preheat (REFERENCE_PREHEAT_PREFIXES).
        # use_fddt_only_n_epochs (base.yaml:59) is the reference's second phase condition (trainers.py:122: epoch >= n_epochs
        # AND global_step >= n_steps); without a data loader an epoch is `steps_per_epoch` optimizer steps.
        if use_fddt_only_n_epochs:
            if not steps_per_epoch:
                raise ValueError("use_fddt_only_n_epochs needs steps_per_epoch (optimizer steps in one pass over the training set)")
            use_fddt_only_n_steps = max(int(use_fddt_only_n_steps), int(use_fddt_only_n_epochs) * int(steps_per_epoch))
        # graph=True: after one eager step per phase the whole step (zero-grad, forward, backward, clip, AdamW, bf16 weight
        # refresh) is captured in ONE hipGraph per phase and replayed: small configurations (whisper-base, B = 8: ~1000 launches
        # of a few microseconds each) are bound by the host's launch rate, not by the GPU.  The step has no host-side data
        # dependence: the clip coefficient stays on the device, the schedule's scalars are rewritten in device memory
        # before each replay (FusedAdamW.advance).
        self.first_writer = True      # encoder weight-matrix gradients are written, not accumulated into a zero fill (FlatStore.zero_grad)
        self.graph = bool(graph)
        # (phase, batch signature) -> (CUDAGraph, static batch, static loss), least recently used first.  Real batches vary in
        # label length (and SE-DiCoW enrollment length), and the hard loss is a mean over ALL label positions
        # (modeling_dicow.py:310-323), so labels cannot be padded to length buckets without changing the loss: instead the cache
        # is bounded (max_graphs, LRU eviction), every capture allocates from ONE shared pool (the footprint is the largest
        # graph's, not the sum), and a signature is only captured the second time it is seen.
        self._graphs = collections.OrderedDict()
        self.max_graphs = 8
        self._graph_pool = None
        self._eager_done = set()
        self.augmenter = augmenter          # augment.BatchAugmenter: the collator's training-time block, on the GPU
        freeze_by_keyword(model, frozen_keywords)
        self.store = FlatStore(model, preheat_prefixes)
        self.opt = FusedAdamW(self.store, lr, fddt_lr_multiplier, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                              warmup_steps=warmup_steps, max_steps=max_steps)
        self.reducer = GradReducer(self.store, process_group)
        # Like DistributedDataParallel's constructor (the reference's DP wrapper): every rank starts from rank 0's parameters
        # and buffers ("broadcast"), or the ranks prove they already agree ("verify": checksums, raises on a mismatch).
        self.replica_sync = replica_sync
        self.replica_sync_bytes = sync_replicas(model, self.store, process_group, mode=replica_sync)
        if self.replica_sync_bytes and replica_sync == "broadcast":      # every bf16 compute copy is stale now, frozen decoder included
            model.model.encoder._sig = None
            model.model.encoder._ctc_sig = None
            model._sig = None
        model.model.encoder._segment_hook = self.reducer.segment_ready
        model._segment_hook = self.reducer.segment_ready
        # Staged freezing (reference train.py:174-178 + trainers.py:122-137, dicow_v3.yaml:68 use_fddt_only_n_steps 2000):
        # while fewer than n optimizer steps have been taken only the preheat-prefixed parameters train; the engine skips
        # every weight-gradient GEMM of a parameter that does not require grad, dgrad still flows down to layer 0's FDDT.
        self.use_fddt_only_n_steps = use_fddt_only_n_steps
        self.warmup_phase = use_fddt_only_n_steps > 0
        names = {id(p): n for n, p in model.named_parameters()}
        self._preheat_is_vectors_only = all(p.dim() == 1 and "fddt" in names[id(p)] for p, _, _, pre in self.store.entries if pre)
        if self.warmup_phase:
            self._set_phase(preheat_only=True)

    def _set_phase(self, preheat_only):
        for p, _, _, pre in self.store.entries:
            p.requires_grad_(pre or not preheat_only)
        self.reducer.preheat_only = preheat_only
        self.model.model.encoder._sig = None
        self.model._sig = None

    @property
    def global_step(self):
        return self.opt.t

    def begin_step(self):
        """trainers.py:122-137: once n optimizer steps are done, unfreeze everything the keywords do not freeze."""
        if self.warmup_phase and self.global_step >= self.use_fddt_only_n_steps:
            self._set_phase(preheat_only=False)
            self.warmup_phase = False
        self.store.zero_grad(first_writer=self.first_writer and not self.warmup_phase)

    def finish_step(self):
        """Gradients are in the flat store: exchange (DP), clip, AdamW, invalidate the bf16 compute copies."""
        with tracing.range("exchange"):               # (the buckets themselves leave during "backward", on the side stream)
            self.reducer.finish()
        # A flagged matrix no GEMM wrote still holds the previous step's values.  Under data parallelism its bucket settled it before
        # leaving (GradReducer.segment_ready -> FlatStore.settle_range: zeros on the wire, rank-consistent whatever each rank skipped);
        # what is left here are the matrices of a single-rank run and of segments that were never announced.  After reducer.finish():
        # a fill on the compute stream must not race with an in-place all-reduce still in flight.  A gradient written into a flagged
        # matrix by anything other than the engine's weight-gradient GEMM must clear ``p._grad_overwrite`` itself, or it is zeroed.
        if self.first_writer and not self.warmup_phase:
            self.store.settle_first_writers()
        with tracing.range("optimizer"):
            self.opt.step(preheat_only=self.warmup_phase)
        # the fused optimizer writes through raw pointers (no torch version bump): invalidate the bf16 weight copies --
        # except in the preheat phase when only diagonal / bias FDDT vectors moved (the row kernels read those in fp32)
        enc = self.model.model.encoder
        if not (self.warmup_phase and self._preheat_is_vectors_only):
            enc._sig = None
            enc._ctc_sig = None
        if any(p.requires_grad for p in self.model.model.decoder.parameters()):
            self.model._sig = None

    def _micro(self, batch, scale):
        if self.split_streams and self._can_split(batch):
            return self._micro_split(batch, scale)
        if self.augmenter is not None:      # enrollments are collated "nested" and stay clean (collators.py:189,216-220)
            batch = self.augmenter(dict(batch))
        with tracing.range("forward"):
            out = self.model(**batch)
        with tracing.range("backward"):
            (out.loss if scale == 1.0 else out.loss * scale).backward()
        return out.loss.detach()

    # ---- one batch as two half batches on two streams
    def _can_split(self, batch):
        cfg = self.model.config
        lab = batch.get("labels")
        return (self.split_streams and self.augmenter is None and cfg.ctc_weight == 0.0 and batch.get("enrollments") is None and
                lab is not None and lab.is_cuda and lab.shape[0] >= 2 and lab.shape[0] % 2 == 0 and
                all(not torch.is_tensor(v) or v.dim() == 0 or v.shape[0] == lab.shape[0] for v in batch.values()))

    def _micro_split(self, batch, scale):
        model, enc, sync = self.model, self.model.model.encoder, self._split_sync
        cur = torch.cuda.current_stream()
        if self._split_side is None:                    # (priority streams get hardware queues of their own: profiles/r06_streams.txt)
            pf, pl = (int(x) for x in os.environ.get("DICOW_SPLIT_PRIO", "-1,-1").split(","))
            self._split_side = (torch.cuda.Stream(priority=pf), torch.cuda.Stream(priority=pl))
        s_f, s_l = self._split_side
        hb = batch["labels"].shape[0] // 2
        halves = [{k: (v[i * hb:(i + 1) * hb] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batch.items()} for i in (0, 1)]
        # the loss is the mean over ALL label positions of the batch (modeling_dicow.py:310-323) = the halves' means weighted by their counts
        # -- and the reference normalises in two ways: the hard-label loss is `.mean()` over ALL positions, padding included
        # (modeling_dicow.py:312-321), the soft-label loss divides by the number of non-padding positions (modeling_dicow.py:135-145)
        if getattr(model, "_ts_tables", None) is not None:
            cnt = torch.stack([(h["labels"] != -100).sum() for h in halves]).to(torch.float32)
            wts = cnt / cnt.sum().clamp_min(1.0) * scale
        else:
            wts = torch.full((2,), 0.5 * scale, dtype=torch.float32, device=batch["labels"].device)
        enc._engine()                                   # stale bf16 weight copies are re-cast HERE, on the current stream, before the fork
        model._engine()
        sync.reset()
        enc._split_sync = model._split_sync = sync
        try:
            with tracing.range("forward"):
                outs = [None, None]
                for i, st in (((1, s_l), (0, s_f)) if _SPLIT_LEAD_FIRST else ((0, s_f), (1, s_l))):     # the leader's forward is enqueued first: it stays ahead
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        outs[i] = model(**halves[i])
            with tracing.range("backward"):
                hold = self.reducer.hold
                self.reducer.hold = True                # the leader's segments are half the gradient: nothing leaves yet
                sync.role = "lead"
                with torch.cuda.stream(s_l):
                    (outs[1].loss * wts[1]).backward()
                self.reducer.hold = hold
                sync.role = "follow"
                with torch.cuda.stream(s_f):
                    (outs[0].loss * wts[0]).backward()
        finally:
            sync.role = None
            enc._split_sync = model._split_sync = None
        cur.wait_stream(s_l)
        cur.wait_stream(s_f)
        for o in outs:
            o.loss.record_stream(cur)
        return (outs[0].loss.detach() * wts[0] + outs[1].loss.detach() * wts[1]) / scale

    # ---- whole-step hipGraph
    @staticmethod
    def _sig_of(batch):
        out = []
        for k in sorted(batch):
            v = batch[k]
            out.append((k, TrainStep._sig_of(v)) if isinstance(v, dict) else (k, tuple(v.shape), str(v.dtype)))
        return tuple(out)

    @staticmethod
    def _clone(batch):
        return {k: (TrainStep._clone(v) if isinstance(v, dict) else v.clone()) for k, v in batch.items()}

    @staticmethod
    def _copy_into(dst, src):
        for k, v in src.items():
            if isinstance(v, dict):
                TrainStep._copy_into(dst[k], v)
            else:
                dst[k].copy_(v, non_blocking=True)

    def _invalidate_weight_copies(self):
        enc = self.model.model.encoder
        enc._sig = None
        enc._ctc_sig = None
        if any(p.requires_grad for p in self.model.model.decoder.parameters()):
            self.model._sig = None

    def _graph_step(self, batch):
        if self.reducer.world > 1 or self.reducer.force:
            raise NotImplementedError("TrainStep(graph=True) captures the single-GPU step; the bucketed side-stream exchange is not captured")
        if self.model.config.ctc_weight > 0.0 or self.augmenter is not None:
            raise NotImplementedError("TrainStep(graph=True): the CTC label preparation / the augmentation planner read device data on "
                                      "the host; their steps cannot be captured")
        if self.warmup_phase and self.global_step >= self.use_fddt_only_n_steps:
            self._set_phase(preheat_only=False)
            self.warmup_phase = False
        key = (self.warmup_phase, self._sig_of(batch))
        if key not in self._eager_done:                 # first step of a phase runs eagerly (allocator, one-time setup)
            self._eager_done.add(key)
            return self._eager_step([batch])
        if key not in self._graphs:
            while len(self._graphs) >= max(1, self.max_graphs):     # bounded cache: drop the least recently replayed graph
                self._graphs.popitem(last=False)
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            static = self._clone(batch)
            self._invalidate_weight_copies()            # the capture must contain the bf16 weight refresh
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool):
                self.store.zero_grad(first_writer=self.first_writer and not self.warmup_phase)
                out = self.model(**static)
                out.loss.backward()
                self.opt.launch(preheat_only=self.warmup_phase)
                loss = out.loss.detach()
            self._graphs[key] = (g, static, loss)
            # (capturing does not execute: the replay below is this step)
        self._graphs.move_to_end(key)
        g, static, loss = self._graphs[key]
        self._copy_into(static, batch)
        self.opt.advance(preheat_only=self.warmup_phase)
        g.replay()
        self._invalidate_weight_copies()                # an eager step that follows must re-cast, too
        return loss.clone()

    def step(self, batch, eager=False):
        """One optimizer step on one batch, or -- given a list / tuple of batches -- on their accumulated gradients
        (HF Trainer gradient_accumulation_steps: each micro-batch's mean loss is divided by the number of micro-batches,
        gradients sum in place in the flat store, ranks exchange them once, after the last micro-batch's backward).
        With graph=True a single batch replays the captured step (`eager=True` forces the launch-by-launch path)."""
        micro = list(batch) if isinstance(batch, (list, tuple)) else [batch]
        if self.graph and not eager and len(micro) == 1:
            return self._graph_step(micro[0])
        return self._eager_step(micro)

    def _eager_step(self, micro):
        self.begin_step()
        loss = None
        for i, b in enumerate(micro):
            self.reducer.hold = i + 1 < len(micro)
            l = self._micro(b, 1.0 / len(micro))
            loss = l if loss is None else loss + l
        self.finish_step()
        return loss / len(micro)

    def logged_loss(self, loss):
        """The number the reference's trainer logs: the mean of the ranks' step losses (HF Trainer gathers the per-rank loss and
        averages it).  One scalar all-reduce on the compute stream, only when called (logging_steps); identity at world 1."""
        r = self.reducer
        if r.world == 1:
            return loss
        t = loss.detach().clone().float().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=r.pg)
        return (t / r.world)[0]

    # ---- checkpoint / resume of the optimizer side (the model side is model.state_dict(), reference key names)
    def state_dict(self):
        o = self.opt
        return {"global_step": o.t, "run_t": list(o.run_t), "warmup_phase": self.warmup_phase,
                "exp_avg": self.store.exp_avg.clone(), "exp_avg_sq": self.store.exp_avg_sq.clone(),
                "layout": [(a, b, pre) for a, b, pre in self.store.runs], "entries": self.store.fingerprint()}

    def resync(self):
        """Re-run the start-up replica sync (rank 0's parameters, frozen parameters and buffers broadcast, or verified) -- call it
        after ``model.load_state_dict(...)`` when that happens AFTER this TrainStep was constructed (the resume order
        INTEGRATION.md documents): the constructor's sync has seen the pre-load weights only."""
        if self.replica_sync == "none" or self.reducer.world <= 1:
            return 0
        n = sync_replicas(self.model, self.store, self.reducer.pg, mode=self.replica_sync)
        self.model.model.encoder._sig = None          # the bf16 compute copies follow the (possibly replaced) masters
        self.model.model.encoder._ctc_sig = None
        self.model._sig = None
        return n

    def _sync_host_counters_phase(self, phase):
        keep = self.warmup_phase
        self.warmup_phase = phase
        try:
            return self._sync_host_counters()
        finally:
            self.warmup_phase = keep

    def _sync_host_counters(self):
        """(t, warmup_phase, run_t...) are host mirrors: ranks that resumed from different files would switch phase / count steps
        differently and their per-segment collectives would stop matching.  broadcast: rank 0's values; verify: raise on a mismatch."""
        if self.replica_sync == "none" or self.reducer.world <= 1 or not dist.is_initialized():
            return
        dev = self.store.params.device
        mine = torch.tensor([int(self.opt.t), int(bool(self.warmup_phase))] + [int(x) for x in self.opt.run_t], dtype=torch.int64, device=dev)
        if self.replica_sync == "verify":
            got = [torch.zeros_like(mine) for _ in range(self.reducer.world)]
            dist.all_gather(got, mine, group=self.reducer.pg)
            bad = [r for r, g in enumerate(got) if not torch.equal(g, got[0])]
            if bad:
                raise RuntimeError(f"data-parallel replicas resumed with different step counters: ranks {bad} differ from rank 0 "
                                   f"(global_step / warmup_phase / per-run steps {[g.tolist()[:3] for g in got]})")
            return
        src = dist.get_global_rank(self.reducer.pg, 0) if self.reducer.pg is not None else 0
        dist.broadcast(mine, src=src, group=self.reducer.pg)
        vals = mine.tolist()
        self.opt.t, self.opt.run_t = int(vals[0]), [int(x) for x in vals[2:]]
        self.opt.sync_counters()
        return bool(vals[1])

    def load_state_dict(self, sd, allow_legacy_layout=False, sync_model=True):
        """Optimizer state (moments, step counters, phase) of ``state_dict()``.  UNDER DATA PARALLELISM THIS IS A COLLECTIVE
        (``replica_sync`` != "none"): EVERY rank must call it, each after loading its model weights; a call on rank 0 alone hangs in
        the broadcast.  With ``sync_model=True`` (default) the whole model state then follows rank 0 -- the flat parameter store, the
        frozen parameters (the 3.6 GB decoder of large-v3) and the buffers -- so a rank that loaded only the optimizer state has
        its freshly loaded WEIGHTS overwritten by rank 0's; a checksum mismatch before the broadcast is logged as a warning.
        ``sync_model=False``: only the moments and counters are synchronised (every rank is trusted to have loaded the same
        weights; ``replica_sync="verify"`` still checks them)."""
        if [tuple(r) for r in sd["layout"]] != [tuple(r) for r in self.store.runs]:
            raise ValueError("optimizer state was saved for a different set of trainable parameters")
        # the runs only fix the run BOUNDARIES; the order of the parameters inside them is part of the layout too (it changed in
        # round 3: weight matrices before vectors) -- moments saved under another order would land on the wrong parameters
        if "entries" not in sd:
            if not allow_legacy_layout:
                raise ValueError("optimizer state has no per-parameter layout fingerprint (saved by an older version): the order of the "
                                 "parameters inside the flat buffers cannot be checked; pass allow_legacy_layout=True if it was saved by "
                                 "a version with today's parameter order (weight matrices before vectors inside a layer), or re-save it")
            log.warning("optimizer state without a layout fingerprint accepted on request: only the run boundaries were checked")
        elif [tuple(e) for e in sd["entries"]] != self.store.fingerprint():
            mine = {e[0]: e for e in self.store.fingerprint()}
            diff = [e[0] for e in sd["entries"] if tuple(e) != mine.get(e[0])][:4]
            raise ValueError(f"optimizer state was saved under a different flat-buffer layout (first differing parameters: {diff})")
        self.opt.t, self.opt.run_t = int(sd["global_step"]), list(sd["run_t"])
        self.opt.sync_counters()
        self.store.exp_avg.copy_(sd["exp_avg"])
        self.store.exp_avg_sq.copy_(sd["exp_avg_sq"])
        phase = bool(sd["warmup_phase"])
        if self.replica_sync != "none" and self.reducer.world > 1:
            # resume: parameters (flat store, frozen ones, buffers), moments and device counters follow rank 0 (or are checked), and
            # so do the host mirrors of the step counters -- ranks that read different files must not diverge in phase / step
            if sync_model or self.replica_sync == "verify":
                if self.replica_sync == "broadcast":
                    try:                                  # say so when the broadcast is about to overwrite weights that differ
                        sync_replicas(self.model, self.store, self.reducer.pg, mode="verify")
                    except RuntimeError as ex:
                        log.warning("load_state_dict: the ranks' model weights differ before the resume broadcast -- rank 0's replace the others' (%s)", ex)
                sync_replicas(self.model, self.store, self.reducer.pg, mode=self.replica_sync,
                              extra=(self.store.exp_avg, self.store.exp_avg_sq, self.opt.counters))
            else:
                sync_replicas(torch.nn.Module(), None, self.reducer.pg, mode=self.replica_sync,
                              extra=(self.store.exp_avg, self.store.exp_avg_sq, self.opt.counters))
            p0 = self._sync_host_counters_phase(phase)
            phase = p0 if p0 is not None else phase
            self.model.model.encoder._sig = None
            self.model.model.encoder._ctc_sig = None
            self.model._sig = None
        sd = dict(sd, warmup_phase=phase)
        if bool(sd["warmup_phase"]) != self.warmup_phase:
            self._set_phase(preheat_only=bool(sd["warmup_phase"]))
            self.warmup_phase = bool(sd["warmup_phase"])
