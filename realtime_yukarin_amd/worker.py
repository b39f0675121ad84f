"""Host-side mirror of the convert worker loop (/root/reference/realtime_voice_conversion/worker/convert_worker.py:17-59)
for SURVEY.md section 8(f) row 3: same arguments, same `Item` in / `Item` out protocol and the same stream arithmetic
(`stream.add` at `extra_time + k * time_length`, `StreamWrapper.process_next`), built on the reference's own
`ConvertStream` / `StreamWrapper` classes (imported from the maintainer's installed `realtime_voice_conversion` package
at call time) and on this package's `VoiceChanger`.  Two differences, both on the host side of the hot path:

* segments that no later window can fetch are dropped after every window (`BaseStream.remove`, base_stream.py:32-33, is
  never called by the reference's loops, so its `stream.stream` list and every `fetch` scan grow for as long as the
  process lives);
* the queues may be `multiprocessing.Queue` or `transport.FeatureQueue`; an item of `None` ends the loop (the reference
  loop has no exit and is killed with its parent)."""
import logging
import time

from .voice_changer import VoiceChanger


def retire_time(current_time: float, time_length: float, extra_time: float) -> float:
    """Latest `end_time` that is safe to pass to `stream.remove` once the wrapper's clock reads `current_time`: the next
    window fetches [current_time - extra_time, current_time + time_length + extra_time] (stream_wrapper.py:11-18,
    base_stream.py:41-43); one more window of slack absorbs the float accumulation of the two clocks."""
    return current_time - extra_time - time_length


def convert_worker(acoustic_converter, super_resolution, time_length: float, extra_time: float, input_silent_threshold: float,
                   queue_input, queue_output, acquired_lock) -> None:
    from realtime_voice_conversion.stream import ConvertStream, StreamWrapper
    logger = logging.getLogger('convert')
    stream = ConvertStream(voice_changer=VoiceChanger(super_resolution=super_resolution, acoustic_converter=acoustic_converter,
                                                      threshold=input_silent_threshold))
    stream_wrapper = StreamWrapper(stream=stream, extra_time=extra_time)
    acquired_lock.release()
    start_time = extra_time
    while True:
        item = queue_input.get()
        if item is None:
            return
        start = time.time()
        stream.add(start_time=start_time, data=item.item)
        start_time += time_length
        item.item = stream_wrapper.process_next(time_length=time_length)
        queue_output.put(item)
        stream.remove(end_time=retire_time(stream_wrapper._current_time, time_length, extra_time))
        logger.debug('%s: %s', item.index, time.time() - start)
