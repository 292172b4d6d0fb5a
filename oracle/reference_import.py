"""ORACLE — TEST INFRASTRUCTURE ONLY.  Imports the REAL reference modules from /root/reference (build container
only; the GPU box has no /root/reference) so the restatement in oracle/stylegan2_ref.py can be validated and golden
vectors generated.  Nothing is copied: the reference is imported in place, with three stub module trees for
dependencies that are absent here:
  my_utils.graph_writer.graph_writer        needs wrapt/pyvis      -> pass-through CallWrapper, no-op ModuleSpace/draw
  my_utils.photometric_optimization[...]    empty git submodule    -> empty modules
"""
import contextlib
import io
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def _stub(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install_stubs():
    if "my_utils.graph_writer.graph_writer" in sys.modules and getattr(
            sys.modules["my_utils.graph_writer.graph_writer"], "_gif_stub", False):
        return
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    gw_pkg = _stub("my_utils.graph_writer")
    gw = _stub("my_utils.graph_writer.graph_writer")
    gw._gif_stub = True

    class ModuleSpace:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    gw.ModuleSpace = ModuleSpace
    gw.CallWrapper = lambda module, node_tracing_name=None: module
    gw.draw = lambda *a, **k: None
    gw_pkg.graph_writer = gw
    import my_utils  # real package from the reference
    my_utils.graph_writer = gw_pkg
    po = _stub("my_utils.photometric_optimization")
    for sub in ("models", "gif_helper", "util"):
        m = _stub(f"my_utils.photometric_optimization.{sub}")
        setattr(po, sub, m)
    flame = _stub("my_utils.photometric_optimization.models.FLAME")
    sys.modules["my_utils.photometric_optimization.models"].FLAME = flame
    my_utils.photometric_optimization = po


def reference_modules():
    """Returns (StyledGenerator, Discriminator, common_layers module) of the real reference."""
    install_stubs()
    cwd = os.getcwd()
    try:
        os.chdir(REF_ROOT)  # the reference does sys.path.append('../') and relative imports of constants
        with contextlib.redirect_stdout(io.StringIO()):
            from model import stg2_generator, stg2_discriminator, stylegan2_common_layers
    finally:
        os.chdir(cwd)
    return stg2_generator.StyledGenerator, stg2_discriminator.Discriminator, stylegan2_common_layers


def reference_fast_image_reshape():
    """The real dataset_loaders.fast_image_reshape (dataset_loaders.py:26-34).  dataset_loaders imports torchvision and lmdb
    at module level (absent here, unused by this function): two more empty stub modules."""
    install_stubs()
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "lmdb"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    cwd = os.getcwd()
    try:
        os.chdir(REF_ROOT)
        with contextlib.redirect_stdout(io.StringIO()):
            import dataset_loaders
    finally:
        os.chdir(cwd)
    return dataset_loaders.fast_image_reshape


def reference_losses():
    """The real loss_functions.losses module (grad_penalty_loss :87-99, PathLengthRegularizor :102-124,
    InterpolatedTextureLoss.pairwise_texture_loss :147-160 ...).  Importable once dataset_loaders' absent dependencies
    are stubbed; classes whose constructors read licensed FLAME files are used through their unbound methods only."""
    reference_fast_image_reshape()
    cwd = os.getcwd()
    try:
        os.chdir(REF_ROOT)
        with contextlib.redirect_stdout(io.StringIO()):
            from loss_functions import losses
    finally:
        os.chdir(cwd)
    return losses
