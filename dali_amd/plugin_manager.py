"""Loading of out-of-tree operator libraries (reference: dali/python/nvidia/dali/plugin_manager.py:19-71,
dali/plugin/plugin_manager.cc:26-112).

A plug-in is a shared library compiled against dali_amd/host/framework.h whose static initialisers register schemas
(DALI_SCHEMA) and operator factories (DALI_REGISTER_OPERATOR).  After the library is loaded the `fn` / `ops`
namespaces are regenerated, so `fn.my_namespace.my_op` exists exactly as for a built-in operator."""
import os

from . import _backend as _b


def _reload():
    from . import fn, ops
    _b._schema_cache.clear()
    fn._populate()
    ops._populate()


def load_library(library_path, global_symbols=False):
    """Loads a plug-in containing one or more operators.  Raises RuntimeError when the library cannot be loaded."""
    lib = _b._lib()
    _b.check(lib.daliamdLoadLibrary(str(library_path).encode(), 1 if global_symbols else 0))
    _reload()


def load_directory(plugin_dir_path, global_symbols=False):
    """Loads every `libdali_*.so` found (recursively) under `plugin_dir_path`
    ({plugin_dir_path}/{sub_path}/libdali_{plugin_name}.so, like the reference)."""
    if not os.path.isdir(plugin_dir_path):
        return
    for root, _dirs, files in sorted(os.walk(plugin_dir_path)):
        for f in sorted(files):
            if f.startswith("libdali_") and f.endswith(".so"):
                load_library(os.path.join(root, f), global_symbols)


def load_preload_plugins():
    """DALI_PRELOAD_PLUGINS="path1:path2": libraries / directories loaded at import (plugin_manager.cc:85-112)."""
    for p in filter(None, os.environ.get("DALI_PRELOAD_PLUGINS", "").split(":")):
        (load_directory if os.path.isdir(p) else load_library)(p)
