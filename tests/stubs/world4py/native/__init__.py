class _NA(object):
    def __getattr__(self, name):
        raise NotImplementedError('world4py stub (tests/stubs): WORLD synthesis is outside the accelerated path')


structures = _NA()
apidefinitions = _NA()
utils = _NA()
