def _na(*a, **k):
    raise NotImplementedError('pyworld stub (tests/stubs): WORLD is outside the accelerated path')


harvest = dio = stonemask = cheaptrick = d4c = synthesize = code_aperiodicity = decode_aperiodicity = _na


def get_cheaptrick_fft_size(fs, f0_floor=71.0):
    import math
    return int(2 ** (1 + int(math.log2(3.0 * fs / f0_floor + 1))))
