"""Import-compatibility stubs for `gsplat.strategy`.

MTGS does not use gsplat's densification strategies (it has its own refinement code in
gaussian_model/vanilla_gaussian_splatting.py:476-577), but nerfstudio's built-in splatfacto model --
imported when nerfstudio registers its methods -- may import these names from gsplat at module load
([NS-RECALL]).  The stubs keep that import from failing; using them raises."""


class _Unavailable:
    _name = "strategy"

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __getattr__(self, item):
        raise NotImplementedError(
            f"gsplat.strategy.{self._name} is not implemented by the MI355X drop-in (MTGS does not use it)")


class DefaultStrategy(_Unavailable):
    _name = "DefaultStrategy"


class MCMCStrategy(_Unavailable):
    _name = "MCMCStrategy"


__all__ = ["DefaultStrategy", "MCMCStrategy"]
