"""Import shim used ONLY by oracle/gen_golden.py (test infrastructure, runs in the build container).

The reference (/root/reference, read-only) has import-time dependencies that are not installed here
(omegaconf, diffusers, loguru, ftfy, imageio, boto3 ...).  None of them is executed on the EDM path;
they only have to satisfy `import` statements.  This module injects inert stand-in *Python modules*
for those names into sys.modules and then imports the reference.  The arithmetic that runs is 100 %
reference code + torch.  Nothing from here (or from the reference) ships to the GPU box.
"""
import sys
import types
import importlib.machinery

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Inert object: any attribute access / call returns another inert object."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        obj = type(name, (_Anything,), {})
        setattr(self, name, obj)
        return obj


def _stub(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _StubModule(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package so `import a.b` works
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_stub(parent), child, m)
    return m


def install_stubs():
    # omegaconf: DictConfig must behave like a dict with attribute access (LazyCall builds them)
    om = _stub("omegaconf")

    class DictConfig(dict):
        def __init__(self, content=None, flags=None, **kw):
            super().__init__(content or {})

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    class ListConfig(list):
        def __init__(self, content=None, flags=None, **kw):
            super().__init__(content or [])

    class OmegaConf:
        @staticmethod
        def to_container(x, **kw):
            return x

        @staticmethod
        def create(x=None, **kw):
            return DictConfig(x) if isinstance(x, dict) or x is None else x

    om.DictConfig, om.ListConfig, om.OmegaConf = DictConfig, ListConfig, OmegaConf

    lg = _stub("loguru")

    class _Logger:
        def __getattr__(self, name):
            def f(*a, **k):
                return self if name in ("opt", "bind") else 0

            return f

    lg.logger = _Logger()
    for name in (
        "diffusers", "diffusers.models", "diffusers.utils", "ftfy", "imageio", "imageio.v3", "boto3", "botocore",
        "botocore.exceptions", "botocore.config", "wandb", "webdataset", "av", "cv2", "pynvml", "timm", "hydra",
        "hydra.core", "hydra.core.global_hydra", "attrs",
    ):
        try:
            __import__(name)
        except Exception:
            _stub(name)


def import_reference():
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import fastgen.networks.EDM.network as edm_net  # noqa
    import fastgen.networks.noise_schedule as ns  # noqa
    import fastgen.methods.model as model  # noqa
    return edm_net, ns, model
