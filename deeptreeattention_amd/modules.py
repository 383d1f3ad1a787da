"""Stand-alone forwards of the reference's building blocks (conv_module, attention modules, Classifier).

Filled in by the module-level C-ABI entry points; see Hang2020.py for the network-level path."""


def _todo(name):
    raise NotImplementedError(
        f"{name}: stand-alone forward is not wired to the HIP library yet; use it inside "
        "spectral_network / spatial_network / Hang2020 / vanilla_CNN")


def conv_module_forward(mod, x, pool):
    _todo("conv_module")


def attention_forward(mod, x, kind):
    _todo(kind + "_attention")


def classifier_forward(mod, features):
    _todo("Classifier")
