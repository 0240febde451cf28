"""Import shims that let the reference's own LoRA / scheduler modules execute in this container (no diffusers,
optimum, torchao, torchaudio, av installed).  Used ONLY by make_golden.py, which runs here (where /root/reference is
mounted) and writes the fixtures next to this file; nothing at test/bench time imports the reference."""
import importlib.machinery
import sys
import types

REFERENCE = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, item):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


class _StubMeta(type):
    """metaclass of the stub classes: class attributes that do not exist are stub classes too (transforms.ColorJitter, Enum-like members)"""

    def __iter__(cls):
        return iter(())

    def __getattr__(cls, item):
        if item.startswith("__"):
            raise AttributeError(item)
        sub = _StubMeta(item, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None,
                                   "__getattr__": lambda self, k: _Anything(), "__bool__": lambda self: False})
        setattr(cls, item, sub)
        return sub


class _AutoStub(types.ModuleType):
    """module whose every missing attribute is a fresh stub class (so `from x import Y` and isinstance() work)."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        sub = sys.modules.get(f"{self.__name__}.{item}")
        if sub is not None:
            return sub
        cls = _StubMeta(item, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None,
                                   "__getattr__": lambda self, k: _Anything(), "__bool__": lambda self: False})
        setattr(self, item, cls)
        return cls


def _auto(name):
    m = _AutoStub(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install():
    import transformers  # noqa: F401  (must be imported before the shims shadow anything it probes)

    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    for name in ("av", "diffusers", "diffusers.models", "diffusers.models.attention_processor", "diffusers.models.embeddings",
                 "diffusers.utils", "diffusers.utils.torch_utils", "diffusers.schedulers", "diffusers.configuration_utils",
                 "optimum", "optimum.quanto", "optimum.quanto.tensor", "torchao", "torchao.dtypes", "torchao.quantization",
                 "torchaudio", "lycoris", "lycoris.config", "lycoris.modules", "lycoris.modules.locon", "lycoris.modules.glora",
                 "lycoris.kohya", "lycoris.kohya.utils", "lycoris.wrapper", "lycoris.modules.lokr", "lycoris.modules.base",
                 "lycoris.functional", "lycoris.functional.general", "lycoris.logging", "oyaml", "dotenv", "cv2", "albumentations",
                 "kornia", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "prodigyopt",
                 "bitsandbytes", "peft", "safetensors_rust", "einops_exts", "k_diffusion", "lpips", "open_clip", "timm",
                 "optimum.quanto.quantize", "torchao.quantization.quant_api"):
        if name not in sys.modules:
            _auto(name)
    qp = _auto("torchao.quantization.quant_primitives")
    qp._DTYPE_TO_BIT_WIDTH = {}
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None


class _StubFinder:
    """meta-path finder of last resort for the introspection-only generators (golden_base_model_contract): any not-installed sub-module
    of a package that is already stubbed (diffusers.pipelines.x.y, optimum.quanto.z, ...) and any other missing top-level package
    becomes an _AutoStub, so that the reference's class definitions can be IMPORTED and inspected (never executed)."""

    def __init__(self, roots):
        self.roots = tuple(roots)

    def find_spec(self, name, path=None, target=None):
        top = name.split(".")[0]
        if top in self.roots or isinstance(sys.modules.get(top), _AutoStub):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AutoStub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_stub_finder(extra_roots=()):
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder(extra_roots))
