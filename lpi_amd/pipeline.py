"""Input pipeline of the training loop: batch i+1 is assembled, tokenised and copied to the GPU while batch i trains.

What it stands in for in the reference (methods/sprompt.py:166-167, 297-303): the DataLoader hands the main process a pageable f32 image batch and a
list of caption strings; ``images.cuda()`` copies 154 MB (256 x 3 x 224 x 224 f32) synchronously and ``PromptLearner.forward`` tokenises inside the
step.  At 22 ms per step on an MI355X both would sit on the critical path (a pageable copy alone runs at a few GB/s), so here a producer thread

  1. pulls the next batch from the loader,
  2. gathers its images into a PINNED staging slot (lpi_host_gather: the batch's rows copied once, on several host threads),
  3. tokenises / packs the captions (the plugin's own PromptLearner + engine.PackedIds: host work, C++ BPE),
  4. issues the host-to-device copies on a SIDE stream into a ring of device slots and records an event,

and the training loop only makes its stream wait for that event (no host synchronisation anywhere).  A device slot is reused once the step that read it
has been enqueued and its completion event recorded; a staging slot once its copy has finished.  ``images`` stay f32 as the reference's loader
delivers them (bit-identical input to the f32 patchify kernel) — or uint8 when the dataset hands over decoded pixels (``pixel_format='u8'``: the
loader's ToTensor + Normalize run inside lpi_patchify_u8, bit for bit, and the copy moves a quarter of the bytes).

Python threads are enough: every heavy part (memcpy, BPE, hipMemcpyAsync) runs in native code with the GIL released; the main thread needs ~4 ms of
host time per step to enqueue it (profiles/r04: tools/host_ahead.py).
"""
from __future__ import annotations

import ctypes
import queue
import threading
import time

import torch

from . import _lib


class DeviceBatch:
    """One training batch on the device: ``images`` f32 (or uint8) [B,3,R,R] (a view of a ring slot — valid until the NEXT batch is requested), ``text`` whatever
    ``prepare_text`` returned (device-resident), ``rest`` the loader's remaining fields, ``host_ms`` the producer's per-phase host times."""
    __slots__ = ("images", "text", "rest", "ready", "slot", "host_ms", "h2d", "index")


class BatchPipeline:
    """Iterate ``loader`` with device-resident batches prepared ``depth`` steps ahead.

    loader       any iterable of (images, captions, ...) with images a HOST f32 tensor [B,3,R,R] or a list of B [3,R,R] tensors
                 (utils.data.collate_keep_images), captions a list of strings or a tensor of token ids;
    prepare_text callable(captions) -> object with .to(device) (engine.PackedIds) or a tensor: runs in the producer thread;
    depth        device / staging slots (>= 2);  threads: host threads of the gather.
    """

    def __init__(self, loader, device, prepare_text=None, depth=3, threads=8, timing=False):
        self.loader, self.device = loader, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LpiError("BatchPipeline feeds an MI355X (device must be cuda:N)")
        self.prepare_text = prepare_text
        self.depth = max(2, int(depth))
        self.threads = int(threads)
        self.timing = bool(timing)
        self.side = torch.cuda.Stream(device=self.device)
        self._stage = [None] * self.depth       # pinned host slots
        self._dev = [None] * self.depth         # device slots
        self._copied = [None] * self.depth      # event: H2D of the staging slot finished (recorded on the side stream)
        self._done = [None] * self.depth        # event: the step that read the device slot is complete (recorded on the consumer's stream)
        self._retired = []                      # slot buffers replaced by larger ones
        self._free = threading.Semaphore(self.depth)
        self._q = queue.Queue()
        self._stop = threading.Event()
        self._thread = None
        _lib.load()

    # ------------------------------------------------------------------ producer
    def _slot_buffers(self, slot, shape, dtype):
        st = self._stage[slot]
        if st is None or st.dtype != dtype or tuple(st.shape[1:]) != tuple(shape[1:]) or st.shape[0] < shape[0]:
            if st is not None:      # a step in flight may still read the old device slot: the old pair lives until the pipeline is dropped
                self._retired.append((st, self._dev[slot]))
            st = self._stage[slot] = torch.empty(shape, dtype=dtype, pin_memory=True)
            self._dev[slot] = torch.empty(shape, dtype=dtype, device=self.device)
        return st, self._dev[slot]

    def _gather(self, images, stage):
        """images (list of [3,R,R] tensors, or one [B,3,R,R] tensor; f32, or uint8 pixels for the in-kernel normalisation) -> stage[:B] on
        self.threads host threads."""
        keep = stage.dtype
        if torch.is_tensor(images):
            images = images if images.dtype == keep else images.to(keep)
            rows = [images[i] for i in range(images.shape[0])] if images.is_contiguous() else [t.contiguous() for t in images]
        else:
            rows = [t if (t.dtype == keep and t.is_contiguous()) else t.to(keep).contiguous() for t in images]
        n, each = len(rows), rows[0].numel() * rows[0].element_size()
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in rows])
        rc = _lib.load().lpi_host_gather(stage.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), n, each, self.threads)
        if rc != 0:
            raise _lib.LpiError(f"lpi_host_gather failed with code {rc}")
        return n

    def _produce(self):
        q, free, stop = self._q, self._free, self._stop      # this pass's hand-over state (a later pass replaces the attributes)
        try:
            torch.cuda.set_device(self.device)
            it = iter(self.loader)
            index = 0
            while not stop.is_set():
                t0 = time.perf_counter()
                try:
                    item = next(it)
                except StopIteration:
                    break
                images, captions, rest = item[0], item[1], tuple(item[2:])
                t1 = time.perf_counter()
                while not free.acquire(timeout=0.2):
                    if stop.is_set():
                        return
                slot = index % self.depth
                B = images.shape[0] if torch.is_tensor(images) else len(images)
                one = images[0]
                shape = (B,) + tuple(one.shape)
                if self._copied[slot] is not None:
                    self._copied[slot].synchronize()          # the staging slot's previous copy has left the host buffer
                stage, dev = self._slot_buffers(slot, shape, torch.uint8 if one.dtype == torch.uint8 else torch.float32)
                t2 = time.perf_counter()
                self._gather(images, stage)
                t3 = time.perf_counter()
                text = captions
                if self.prepare_text is not None:
                    text = self.prepare_text(captions if torch.is_tensor(captions) else list(captions))
                t4 = time.perf_counter()
                b = DeviceBatch()
                with torch.cuda.stream(self.side):
                    if self._done[slot] is not None:
                        self.side.wait_event(self._done[slot])   # the step that read this device slot has finished
                    e0 = torch.cuda.Event(enable_timing=True) if self.timing else None
                    if e0 is not None:
                        e0.record(self.side)
                    dev[:B].copy_(stage[:B], non_blocking=True)
                    if torch.is_tensor(text):
                        text = text.pin_memory().to(self.device, non_blocking=True) if not text.is_cuda else text
                    elif hasattr(text, "to"):
                        text = text.to(self.device, non_blocking=True)      # engine.PackedIds: pinned copies, enqueued only
                    cp = torch.cuda.Event(enable_timing=self.timing)
                    cp.record(self.side)
                self._copied[slot] = cp
                b.images, b.text, b.rest, b.ready, b.slot, b.index = dev[:B], text, rest, cp, slot, index
                b.h2d = (e0, cp) if self.timing else None
                b.host_ms = {"loader": 1e3 * (t1 - t0), "slot_wait": 1e3 * (t2 - t1), "gather": 1e3 * (t3 - t2), "tokenise_pack": 1e3 * (t4 - t3),
                             "issue_h2d": 1e3 * (time.perf_counter() - t4)}
                q.put(b)
                index += 1
            q.put(None)
        except BaseException as e:      # noqa: BLE001 — handed to the consumer, which re-raises it
            q.put(e)

    # ------------------------------------------------------------------ consumer
    def __iter__(self):
        if self._thread is not None and self._thread.is_alive():
            raise RuntimeError("this BatchPipeline is still being iterated (one pass over the loader at a time)")
        # a new pass over the loader (the next epoch): the pinned / device slots and their events stay, the hand-over state starts fresh
        self._free = threading.Semaphore(self.depth)
        self._q = queue.Queue()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._produce, name="lpi-batch-pipeline", daemon=True)
        self._thread.start()
        q, free = self._q, self._free
        out = None      # the batch handed out and not yet returned
        try:
            while True:
                b = q.get()
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(b.ready)                       # stream-ordered: the host does not wait
                if hasattr(b.text, "record_stream"):
                    b.text.record_stream(cur)                 # allocated on the side stream, read on this one
                out = b
                yield b
                # the consumer has enqueued the step that read this slot (it is back for the next batch): mark its end, free the slot
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._done[b.slot] = ev
                out = None
                free.release()
        finally:
            if out is not None:
                # the pass ends with a batch still out (the caller broke out of its loop, or its step raised): whatever it enqueued on this slot ends
                # before the event below, so the next pass — slot indices restart at 0 — cannot overwrite a ring slot a step in flight still reads
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._done[out.slot] = ev
            self.close()

    def close(self):
        self._stop.set()
        t = self._thread
        if t is not None and t.is_alive() and t is not threading.current_thread():
            for _ in range(self.depth + 1):     # wake a producer parked on the slot semaphore
                self._free.release()
            t.join(timeout=10.0)
