"""Import shim: `import otmb_amd` loads the package that lives in the directory
`oceantransportmatrixbuilder.jl_amd/` (a dot in a directory name cannot be written in an
import statement).  After this module runs, `sys.modules["otmb_amd"]` is that package and
`import otmb_amd.capi` etc. resolve inside the directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oceantransportmatrixbuilder.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "otmb_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["otmb_amd"] = _mod
_spec.loader.exec_module(_mod)
