"""Host -> HBM staging of the frame stream: the reference moves every batch with a blocking
`in_.to(device)` from pageable memory (src/tta_main.py:519-523), which serialises ~6.8 MB of PCIe traffic per KITTI frame
with the step.  Here frame k+1 is copied into a pinned slot and sent on a dedicated copy stream while the step of frame k
runs.  Slot reuse is ordered by the host waiting on events that are, in steady state, already complete (three slots: the
slot being refilled was consumed two steps ago), which is also the loop's backpressure: the host never runs more than
`slots` frames ahead of the GPU.

The pageable -> pinned copy is a single-threaded numpy copy ON PURPOSE: torch's `Tensor.copy_` fans a 5 MB copy out over
every core it sees (256 on the GPU box) and the spinning OpenMP team exhausts the container's CPU quota (cgroup cpu.max =
16 CPUs) within a few frames -- the process is then throttled for the rest of the 100 ms period, measured as an 80-100 ms
stall every ~5 frames (measured in round 2: 21.8 ms/frame with torch's copy, 2.35 ms/frame with one thread).

    stager.submit(image_k1, sparse_k1)        # host memcpy into pinned slot, async H2D on the copy stream
    image, sparse = stager.acquire()          # host waits for that slot's H2D (submitted a step ago: already there)
    engine.step(image, sparse)                # enqueue only
    stager.release()                          # the slot may be overwritten once the step has consumed it

PyTorch supplies the pinned memory, streams and events (plumbing); no arithmetic happens here.
"""
import numpy as np
import torch


def _as_numpy(x):
    return x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class FrameStager:
    def __init__(self, n, height, width, slots=3, device=None, copy_stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError('FrameStager needs a HIP device (there is no CPU fallback)')
        assert slots >= 2
        self.shape_image, self.shape_sparse = (n, 3, height, width), (n, 1, height, width)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.slots = slots
        self.host = [(torch.empty(self.shape_image, dtype=torch.float32).pin_memory(),
                      torch.empty(self.shape_sparse, dtype=torch.float32).pin_memory()) for _ in range(slots)]
        self.host_np = [(a.numpy(), b.numpy()) for a, b in self.host]          # views of the pinned slots
        self.dev = [(torch.empty(self.shape_image, dtype=torch.float32, device=self.device),
                     torch.empty(self.shape_sparse, dtype=torch.float32, device=self.device)) for _ in range(slots)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if copy_stream is None else copy_stream
        self.ready = [torch.cuda.Event() for _ in range(slots)]          # H2D into the slot has finished
        self.consumed = [torch.cuda.Event() for _ in range(slots)]       # the step that read the slot has finished
        self.head = self.tail = 0                                        # submitted / acquired counters
        self.held = False
        self.bytes_per_frame = 4 * (int(np.prod(self.shape_image)) + int(np.prod(self.shape_sparse)))

    def in_flight(self):
        return self.head - self.tail

    def submit(self, image, sparse):
        """image N x 3 x H x W, sparse N x 1 x H x W: numpy arrays or CPU tensors (any float/uint8 dtype; converted to
        float32 while copying into the pinned slot).  Returns immediately after enqueueing the H2D copy."""
        if self.in_flight() >= self.slots:
            raise RuntimeError('all %d staging slots are in flight: acquire()/release() one first' % self.slots)
        k = self.head % self.slots
        if self.head >= self.slots:
            self.consumed[k].synchronize()         # the step that read this slot (slots frames ago) has finished => so has its H2D
        hi, hs = self.host[k]
        np.copyto(self.host_np[k][0], _as_numpy(image).reshape(self.shape_image), casting='unsafe')
        np.copyto(self.host_np[k][1], _as_numpy(sparse).reshape(self.shape_sparse), casting='unsafe')
        with torch.cuda.stream(self.copy_stream):
            self.dev[k][0].copy_(hi, non_blocking=True)
            self.dev[k][1].copy_(hs, non_blocking=True)
            self.ready[k].record(self.copy_stream)
        self.head += 1

    def acquire(self):
        """Device tensors of the oldest submitted frame, after its copy has landed."""
        if self.held or self.tail >= self.head:
            raise RuntimeError('acquire() without a submitted frame (or before release() of the previous one)')
        k = self.tail % self.slots
        self.ready[k].synchronize()
        self.held = True
        return self.dev[k]

    def peek_next(self, stream=None):
        """Device tensors of the frame AFTER the acquired one (already submitted; None otherwise) for frame pipelining
        (Engine.step(..., next_frame=...)).  No host wait: `stream` (the engine's prefix_stream()) is made to wait for that slot's copy."""
        if not self.held or self.tail + 1 >= self.head:
            return None
        k = (self.tail + 1) % self.slots
        (torch.cuda.current_stream() if stream is None else stream).wait_event(self.ready[k])
        return self.dev[k]

    def release(self):
        """Mark the acquired slot reusable once everything enqueued so far on the current stream has run."""
        if not self.held:
            raise RuntimeError('release() without acquire()')
        self.consumed[self.tail % self.slots].record(torch.cuda.current_stream())
        self.tail += 1
        self.held = False
