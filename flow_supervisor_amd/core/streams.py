"""A second HIP stream for the independent branches of a forward pass.

The feature encoder + volume build and the context encoder of RAFT.forward (pytorch/core/raft.py:99-113) read only the input
frames; so do the uncropped frames' encodings of the flow-supervisor forward (pytorch/core/l2l.py:95-101), which the reference
computes in the middle of the iteration loop.  Issued on two streams they overlap: the tail of one branch's kernels runs beside
the other's, and at one or two pairs per GPU -- where the update block's launches leave half of the CUs without a workgroup --
a whole encoder pass hides behind the student's iterations.  In a captured step the two streams become two branches of the
hipGraph.  Autograd runs every node's backward on the stream its forward ran on and orders the streams itself, so the backward
overlaps the same way.

Measured and not kept: an iteration's GMA Aggregate backward block (attn^T @ dagg, to_v's data gradient; nothing in the recurrence
waits for it) on the second stream (+0.5 %, inside the noise); the mask head's half of the fused 3x3 head convolution on the second stream, off the recurrence's chain
(-0.5 % / -1 %: two 256-output launches cost more than the 512-output one by more than the overlap returns); the feature encoder as two chains of one frame each on two streams (-2 % config 3, -6 % at one pair: the
half-size launches and the second gradient contribution per parameter cost more than the overlap returns); (same A/B script,
WHAT=wgrad at the time) the update block's once-per-step weight gradients on a third
stream beside the encoders' backward (-0.3 % config 3, -1.7 % at one pair per GPU), and the batched ones of them issued early, beside
the recurrence's data-gradient chain (-2 % / -6 %: the big launches take the CUs the serial chain is waiting for).  A high-priority stream for the caller's chain (priority range
on this stack: 0 and -1) is no way to make such background work cheap: the captured step on a priority -1 stream replays in 47.8
instead of 36.6 ms (60.5 instead of 33.5 at one pair).  One thing that
experiment showed is worth keeping in mind for any node placed on another stream: the autograd engine orders a node's stream
behind the producers of the gradients it RECEIVES -- a node whose incoming gradients are never materialised (update._ParamFn's
anchor) is not ordered behind anything and needs its own wait_stream.

Rules the callers keep: weights packed for the kernels (extractor._prepare_packs) are built on the caller's stream BEFORE the
fork when both streams will read them; tensors produced on the side stream are handed to `join`, which makes the caller's stream
wait and tells the caching allocator about their second stream.
"""
import threading

import torch

OVERLAP = True          # False: every branch on the caller's stream, one after the other
_SIDE = {}              # (device index, which, host thread) -> stream
_LOCK = threading.Lock()


def side_stream(device, which=0):
    """The calling THREAD's side stream `which` of `device` (one stream per (device, which) for the whole process made
    two host threads driving the same device share a stream -- and with it the split-K scratch and the zero-pool chunk that
    ops keys on (device, stream): thread B's partial-tile kernel could land between thread A's partial and finish kernels)."""
    device = torch.device(device)
    key = (device.index, which, threading.get_ident())
    s = _SIDE.get(key)
    if s is None:
        with _LOCK:
            s = _SIDE.get(key)
            if s is None:
                s = _SIDE[key] = torch.cuda.Stream(device=device)
    return s


_QUIET_DEPTH = [0]
_QUIET_PREV = [True]


class accumulate_grad_warning_off:
    """Context manager: torch's once-per-process warning about an AccumulateGrad node whose gradient arrives from another stream
    is switched off INSIDE this package's forward / backward only (round 4 switched it off for the whole process).
    A parameter used on both streams (the encoders of the flow-supervisor forward) gets gradients from nodes on two streams; its
    AccumulateGrad node belongs to one of them and the engine synchronises the other -- here that is the design."""

    # (covers the TrainStep / SemiTrainStep backward calls only -- a bare loss.backward() on the package's two-stream
    #  forward still gets torch's warning; the depth counter is guarded, and a user who had switched the warning off keeps it off)
    def __enter__(self):
        setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if setter is not None:
            with _LOCK:
                if _QUIET_DEPTH[0] == 0:
                    getter = getattr(torch._C, "_warn_on_accumulate_grad_stream_mismatch", None)      # (private: present on torch 2.10)
                    try:
                        _QUIET_PREV[0] = bool(getter()) if callable(getter) else True
                    except Exception:       # noqa: BLE001
                        _QUIET_PREV[0] = True
                    setter(False)
                _QUIET_DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if setter is not None:
            with _LOCK:
                _QUIET_DEPTH[0] -= 1
                if _QUIET_DEPTH[0] == 0:
                    setter(_QUIET_PREV[0])
        return False


def fork(device, which=0):
    """-> the side stream, ordered behind everything the caller's stream has been given so far."""
    side = side_stream(device, which)
    side.wait_stream(torch.cuda.current_stream(device))
    return side


def mark(device, which=0):
    """An event behind what the side stream has been given so far (for a `join` that must not wait for later work on it)."""
    ev = torch.cuda.Event()
    ev.record(side_stream(device, which))
    return ev


def join(device, *tensors, event=None, which=0):
    """The caller's stream waits for the side stream (or only up to `event`, see `mark`); `tensors` (allocated there) are used
    on the caller's stream from now on."""
    main = torch.cuda.current_stream(device)
    if event is not None:
        main.wait_event(event)
    else:
        main.wait_stream(side_stream(device, which))
    for t in tensors:
        if t is not None:
            t.record_stream(main)


def _capturing(stream):
    with torch.cuda.stream(stream):
        return torch.cuda.is_current_stream_capturing()


def order_current_behind_all(device, *more):
    """The current stream waits for every side stream of `device` that has work in flight (and for `more`): for code that runs
    inside the backward pass on whatever stream its autograd node has and reads what nodes on OTHER streams produced -- a
    gradient bucket's pack + all-reduce issued from a hook (parallel.FlatGradients): the engine orders a node behind the
    producers of ITS inputs, not behind the other parameters' gradients that share its bucket.
    Which streams: inside a hipGraph capture only the side streams that are part of the capture -- a wait on
    un-captured work is a capture error, and a stream this step never forked cannot hold anything the capture reads; eagerly,
    streams that report idle are skipped (nothing to wait for; two event operations per bucket saved each)."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    capturing = torch.cuda.is_current_stream_capturing()
    # (every thread's side streams of the device: this runs in the autograd engine's worker thread, the forks were made by
    #  the thread that ran the forward; another model's idle / un-captured streams fall out below)
    for (idx, _, _tid), s in list(_SIDE.items()):
        if idx != device.index or s == cur:
            continue
        if capturing:
            if _capturing(s):
                cur.wait_stream(s)
        elif not s.query():
            cur.wait_stream(s)
    for s in more:
        if s is not None and s != cur:
            cur.wait_stream(s)
