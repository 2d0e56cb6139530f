"""Tracker registries — the reference's plug-in point for the EMM head.

Reference: siammot/utils/registry.py:1-4 (``SIAMESE_TRACKER`` / ``TRACKER_SAMPLER``, instances
of [UPSTREAM] ``maskrcnn_benchmark.utils.registry.Registry``) and the lookup in
``build_track_head`` (siammot/modelling/track_head/track_head.py:113-126).

When the reference package is importable (``siammot.utils.registry``) its own registries are
used, so ``siammot_amd.emm.EMM`` lands in the very dict ``build_track_head`` reads; otherwise a
local ``Registry`` with the same ``register`` semantics is provided.
"""


class Registry(dict):
    """dict with ``@registry.register("name")`` decorator support ([UPSTREAM] utils/registry.py)."""

    def register(self, module_name, module=None):
        if module is not None:
            self[module_name] = module
            return module

        def register_fn(fn):
            self[module_name] = fn
            return fn
        return register_fn


def _resolve():
    try:
        from siammot.utils import registry as ref     # the reference, if on sys.path
        return ref.SIAMESE_TRACKER, ref.TRACKER_SAMPLER
    except Exception:
        return Registry(), Registry()


SIAMESE_TRACKER, TRACKER_SAMPLER = _resolve()
