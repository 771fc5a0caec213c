"""Import alias: the package directory is `ms-nets_amd/` (hyphen), which `import` cannot spell."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("ms-nets_amd")
