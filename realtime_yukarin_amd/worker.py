"""Host-side mirror of the convert worker loop (/root/reference/realtime_voice_conversion/worker/convert_worker.py:17-59)
for SURVEY.md section 8(f) row 3: same arguments, same `Item` in / `Item` out protocol and the same stream arithmetic
(`stream.add` at `extra_time + k * time_length`, `StreamWrapper.process_next`), built on the reference's own
`ConvertStream` / `StreamWrapper` classes (imported from the maintainer's installed `realtime_voice_conversion` package
at call time) and on this package's `VoiceChanger`.  Two differences, both on the host side of the hot path:

* segments that no later window can fetch are dropped after every window (`BaseStream.remove`, base_stream.py:32-33, is
  never called by the reference's loops, so its `stream.stream` list and every `fetch` scan grow for as long as the
  process lives);
* the queues may be `multiprocessing.Queue` or `transport.FeatureQueue`; an item of `None` ends the loop (the reference
  loop has no exit and is killed with its parent);
* `ConvertStream.process` (convert_stream.py:32-44) is taken in its two halves -- fetch + queue the window on the GPU, collect + pick --
  so that a backlog of items keeps `depth` windows in flight on the pinned ring of `ry_vc_submit` (copies of one window under the
  kernels of another); a live stream without backlog behaves exactly like the synchronous loop;
* the frames `pick` throws away (the `extra_time` context on either side of the buffer, two thirds of a window in the reference's
  configuration) are announced to the window call (`VoiceChanger.begin(..., discard=(pad, pad))`): stage 2 does not compute them."""
import logging
import time

from .voice_changer import VoiceChanger


def retire_time(current_time: float, time_length: float, extra_time: float) -> float:
    """Latest `end_time` that is safe to pass to `stream.remove` once the wrapper's clock reads `current_time`: the next
    window fetches [current_time - extra_time, current_time + time_length + extra_time] (stream_wrapper.py:11-18,
    base_stream.py:41-43); one more window of slack absorbs the float accumulation of the two clocks."""
    return current_time - extra_time - time_length


def convert_worker(acoustic_converter, super_resolution, time_length: float, extra_time: float, input_silent_threshold: float,
                   queue_input, queue_output, acquired_lock, depth: int = 2) -> None:
    """`depth` windows are kept in flight on the GPU when the input queue holds a backlog (`VoiceChanger.begin` / `finish` over the
    pinned ring of `ry_vc_submit`); with an empty queue every window is finished as soon as it has been queued, so a live stream sees
    the latency of the synchronous loop."""
    import collections
    from realtime_voice_conversion.stream import ConvertStream, StreamWrapper
    logger = logging.getLogger('convert')
    vc = VoiceChanger(super_resolution=super_resolution, acoustic_converter=acoustic_converter, threshold=input_silent_threshold)
    stream = ConvertStream(voice_changer=vc)
    stream_wrapper = StreamWrapper(stream=stream, extra_time=extra_time)
    pad = round(extra_time * stream.in_segment_method.sampling_rate)          # convert_stream.py:40-42
    pending = collections.deque()

    def finish_one():
        item, handle, start = pending.popleft()
        out_feature = vc.finish(handle)
        if pad > 0:
            out_feature = stream.out_segment_method.pick(out_feature, pad, -pad)
        item.item = out_feature
        queue_output.put(item)
        logger.debug('%s: %s', item.index, time.time() - start)

    acquired_lock.release()
    start_time = extra_time
    while True:
        if pending and (len(pending) >= depth or queue_input.empty()):
            finish_one()
            continue
        item = queue_input.get()
        if item is None:
            while pending:
                finish_one()
            return
        start = time.time()
        stream.add(start_time=start_time, data=item.item)
        start_time += time_length
        # ConvertStream.process (convert_stream.py:32-44) in two halves: fetch + queue now, convert result + pick when it is collected
        in_feature = stream.fetch(start_time=stream_wrapper._current_time, time_length=time_length, extra_time=extra_time)
        stream_wrapper._current_time += time_length
        pending.append((item, vc.begin(in_feature, discard=(pad, pad)), start))     # the rows `pick` drops below are not computed by stage 2
        stream.remove(end_time=retire_time(stream_wrapper._current_time, time_length, extra_time))
