"""Import-compatibility stubs for `gsplat.strategy`.

MTGS does not use gsplat's densification strategies (it has its own refinement code in
gaussian_model/vanilla_gaussian_splatting.py:476-577; the device-side equivalent here is
mtgs_amd.densify), but nerfstudio 1.1.5 (/root/reference/requirements.txt:13) imports
`DefaultStrategy` and `MCMCStrategy` from gsplat when it registers its built-in splatfacto method
([NS-RECALL]).  The stubs keep that import -- and config introspection,
copy.deepcopy, pickling, hasattr() on an instance -- from failing; calling a strategy METHOD raises
NotImplementedError by name."""

_METHODS = ("initialize_state", "check_sanity", "step_pre_backward", "step_post_backward")


class _Unavailable:
    _name = "strategy"

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __getattr__(self, item):
        # only reached for names that are not instance / class attributes
        if item in _METHODS:
            raise NotImplementedError(
                f"gsplat.strategy.{self._name}.{item} is not implemented by the MI355X drop-in (MTGS does not use it)")
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {item!r}")


class DefaultStrategy(_Unavailable):
    _name = "DefaultStrategy"


class MCMCStrategy(_Unavailable):
    _name = "MCMCStrategy"


__all__ = ["DefaultStrategy", "MCMCStrategy"]
