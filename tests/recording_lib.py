"""Test helper: the C-ABI library with its launcher calls written down."""


class RecordingLib:
    """The C-ABI library with every launcher call and its return code written down: a test can then assert which fused launchers
    ACCEPTED a shape (a launcher that answers ORCAI_E_UNSUPPORTED makes orcai_amd.training fall back to the separate passes -- silently,
    by design -- so green gradients alone do not say which kernels produced them)."""

    def __init__(self, lib):
        self._lib, self.calls = lib, []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            rc = fn(*args)
            self.calls.append((name, rc, args))
            return rc

        return call

    def rcs(self, name):
        return [rc for n, rc, _ in self.calls if n == name]
