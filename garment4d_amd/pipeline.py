"""Batch executor of the encoder + lbs() hot path: keeps the chip full when the caller produces B = 8 batches one at a time.

The reference's training loop hands the model one batch per iteration (train_temporal.py:239-254) and the encoder folds
(clips, frames) into one batch dimension (modules/mesh_encoder.py:133).  On MI355X a single B = 8 call cannot fill the chip: the
sampling kernels are strictly serial per cloud (1023 + 255 + 63 dependent rounds, one workgroup per cloud = 8 of 256 CUs) and the
shared-MLP launches of 8 clouds are one wave-round each (ramp + tail, no steady state).  Two things fix that, and this class owns both:

  * COALESCING: `coalesce` consecutive steps are gathered into one call on 8 * coalesce clouds (every kernel of the path takes any
    number of clouds; results are bit-identical per cloud -- tests/test_pipeline_gpu.py), so a launch carries many tiles per SIMD and
    the sampling launch costs the same 0.64 ms for 240 clouds as for 8;
  * CONCURRENCY: `streams` such calls are in flight, each a captured hipGraph on its own HIP stream with static input / output
    buffers, so one call's sampling (which leaves most of every CU idle) overlaps another call's shared MLPs.

    pipe = StepPipeline(model, smpl, clouds_per_step=8, n_points=8192, coalesce=30, streams=2)
    fut = pipe.submit(cloud, betas, pose)        # (8, N, 3), (8, 10), (8, 72) HIP tensors; returns at once
    ...
    logits, verts, joints = fut.result()         # waits for the step's call; views into the slot's output buffers

The views stay valid until `streams` further calls have been launched (the slot is then reused); pass copy=True to result() to keep
them.  A partially filled call is launched by flush() (or by result() on one of its steps): the graph then runs on the slot's
stale tail clouds as well -- wasted work, never wrong results.

hipGraph + streams: the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With more streams than queues
the calls serialise; set GPU_MAX_HW_QUEUES >= `streams` in the environment BEFORE the runtime initialises (i.e. before `import
torch`), as bench.py does.  `streams` <= 4 needs nothing.
"""
import ctypes

import torch

from . import _lib
from . import lbs as G
from . import tuning as _tuning


class StepFuture:
    """Result handle of one submitted step."""
    __slots__ = ("_pipe", "_slot", "_j", "_gen")

    def __init__(self, pipe, slot, j, gen):
        self._pipe, self._slot, self._j, self._gen = pipe, slot, j, gen

    def done(self):
        s = self._slot
        return s.launched_gen >= self._gen and s.event.query()

    def result(self, copy=False):
        """(sem_logits (B, N, classes), verts (B, V, 3) | None, joints (B, J, 3) | None) of this step."""
        s = self._slot
        if s.launched_gen < self._gen:
            self._pipe._launch(s)            # its call was still filling up
        if s.launched_gen != self._gen:
            raise RuntimeError("StepFuture.result(): the slot has been reused -- results of a step must be taken (or copied) before "
                               f"{len(self._pipe.slots)} further calls are launched")
        s.event.synchronize()
        b = self._pipe.B
        sl = slice(self._j * b, (self._j + 1) * b)
        out = tuple(None if t is None else t[sl] for t in s.outs)
        return tuple(None if t is None else t.clone() for t in out) if copy else out


class _Slot:
    __slots__ = ("stream", "cloud", "betas", "pose", "graph", "outs", "event", "fill", "gen", "launched_gen")


class StepPipeline:
    def __init__(self, model, smpl=None, clouds_per_step=8, n_points=8192, coalesce=8, streams=2, precision="fp32", pose2rot=True,
                 device=None, use_graph=True, tuning=None, encoder_call=None):
        """model: a Pointnet2MSGSEG in eval mode (its forward_fused is what runs -- or `encoder_call(model, clouds) -> sem_logits`, e.g.
        `lambda m, pc: m(pc)[1]`, the reference's own call form, which in eval + no_grad dispatches to the same kernels and converts the
        features to the reference's (B, C, N) layout); smpl: dict with v_template, shapedirs, posedirs,
        J_regressor, parents, lbs_weights (HIP tensors) or None for the encoder alone.  tuning: the garment4d_amd.tuning.Tuning this
        executor runs under (default: the one in force where it is constructed) -- held for its lifetime, applied around every call it
        launches or captures; two executors in one process can carry different ones."""
        assert coalesce >= 1 and streams >= 1 and not model.training
        self.tuning = tuning if tuning is not None else _tuning.current()
        self.model, self.smpl, self.B, self.N, self.k = model, smpl, int(clouds_per_step), int(n_points), int(coalesce)
        self.precision, self.pose2rot, self.use_graph, self.encoder_call = precision, pose2rot, use_graph, encoder_call
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        self.device = dev
        nb = 0 if smpl is None else smpl["shapedirs"].shape[-1]
        nj = 0 if smpl is None else smpl["J_regressor"].shape[0]
        self.slots = []
        for _ in range(streams):
            s = _Slot()
            s.stream = torch.cuda.Stream(device=dev)
            s.cloud = torch.zeros((self.B * self.k, self.N, 3), dtype=torch.float32, device=dev)
            s.betas = None if smpl is None else torch.zeros((self.B * self.k, nb), dtype=torch.float32, device=dev)
            s.pose = None if smpl is None else (torch.zeros((self.B * self.k, nj * 3), dtype=torch.float32, device=dev) if pose2rot else
                                                torch.eye(3, device=dev).repeat(self.B * self.k, nj, 1, 1).contiguous())
            s.graph, s.outs, s.event = None, None, torch.cuda.Event()
            s.fill, s.gen, s.launched_gen = 0, 1, 0
            self.slots.append(s)
        self._cur = 0
        self._warm()

    # ---- one call = the hot path on a slot's 8 * coalesce clouds
    def _call(self, s):
        with _tuning.use(self.tuning):
            return self._call_tuned(s)

    def _call_tuned(self, s):
        if self.encoder_call is not None:
            from . import fused
            with fused.precision(self.precision):
                logits = self.encoder_call(self.model, s.cloud)
            out = (None, logits)
        else:
            out = self.model.forward_fused(s.cloud, precision=self.precision)
        v = j = None
        if self.smpl is not None:
            P = self.smpl
            v, j = G.lbs(s.betas, s.pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"],
                         pose2rot=self.pose2rot)
        return out[1], v, j

    def _warm(self):
        with torch.no_grad():
            for s in self.slots:                       # eager once: packs weights, sets kernel attributes, sizes the allocator
                with torch.cuda.stream(s.stream):
                    s.outs = self._call(s)
            torch.cuda.synchronize(self.device)
            if self.use_graph:
                for s in self.slots:
                    s.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(s.graph, stream=s.stream):
                        s.outs = self._call(s)
                torch.cuda.synchronize(self.device)

    def _launch(self, s):
        with torch.cuda.stream(s.stream):
            if s.graph is not None:
                s.graph.replay()
            else:
                with torch.no_grad():
                    s.outs = self._call(s)
            s.event.record(s.stream)
        s.launched_gen = s.gen
        s.gen += 1
        s.fill = 0
        if s is self.slots[self._cur]:
            self._cur = (self._cur + 1) % len(self.slots)

    def submit(self, cloud, betas=None, pose=None, inputs_ready=False):
        """Queue one step: cloud (B, N, 3) [+ betas (B, NB), pose (B, J*3) or (B, J, 3, 3)].  Device-to-device copies into the slot's
        buffers on the slot's stream (ordered behind the slot's previous call and -- unless inputs_ready: the inputs are known to be
        complete, e.g. a resident pool -- behind the caller's current stream); launches the call when it is full."""
        s = self.slots[self._cur]
        b = self.B
        sl = slice(s.fill * b, (s.fill + 1) * b)
        if not inputs_ready:
            s.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s.stream):
            # one launch for the step's inputs (g4d_copy_segments_f32): three runtime copies per step are 90 launches per 30-step call
            pairs = [(s.cloud[sl], cloud)] if self.smpl is None else [(s.cloud[sl], cloud), (s.betas[sl], betas), (s.pose[sl], pose)]
            ok = all(src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.numel() == dst.numel() for dst, src in pairs)
            if ok:
                n = len(pairs)
                PA, LA = ctypes.c_void_p * n, ctypes.c_longlong * n
                _lib.call("g4d_copy_segments_f32", n, ctypes.cast(PA(*[d.data_ptr() for d, _ in pairs]), ctypes.c_void_p),
                          ctypes.cast(PA(*[x.data_ptr() for _, x in pairs]), ctypes.c_void_p),
                          ctypes.cast(LA(*[d.numel() for d, _ in pairs]), ctypes.c_void_p), _lib.stream_ptr())
            else:   # host tensors, other dtypes / strides: the runtime's copies
                for dst, src in pairs:
                    dst.copy_(src.reshape(dst.shape), non_blocking=True)
            # The copy is queued behind the slot's previous call (milliseconds); the caller may drop or overwrite its tensors as soon as
            # submit() returns.  Tell the caching allocator that the slot's stream still reads them, so that a freed block is not handed
            # out again (and overwritten on the caller's stream) before the copy has run.  Overwriting a LIVE input in place remains the
            # caller's race, as with any asynchronous copy (wait for the step's future, or hand over a fresh tensor per step).
            for _, src in pairs:
                if src.is_cuda:
                    src.record_stream(s.stream)
        fut = StepFuture(self, s, s.fill, s.gen)
        s.fill += 1
        if s.fill == self.k:
            self._launch(s)
        return fut

    def flush(self):
        """Launch a partially filled call, if any."""
        s = self.slots[self._cur]
        if s.fill:
            self._launch(s)

    def synchronize(self):
        self.flush()
        for s in self.slots:
            s.stream.synchronize()
