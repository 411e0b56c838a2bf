"""Import alias: makes the hyphenated directory `nclt-slam-project_amd/` importable as the
package `nclt_slam_project_amd` (put the repository root on sys.path and `import
nclt_slam_project_amd`)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nclt-slam-project_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
