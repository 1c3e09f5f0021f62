"""Helper of build_ref_variant.sh: build <src tree>'s library as variant <name> and copy it into this tree's lib/."""
import importlib.util
import os
import shutil
import sys

src, repo, name = sys.argv[1:4]
spec = importlib.util.spec_from_file_location("b", os.path.join(src, "oceantransportmatrixbuilder.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
lib = b.build(force=True, name=name)
dst = os.path.join(repo, "oceantransportmatrixbuilder.jl_amd", "lib", os.path.basename(lib))
shutil.copy(lib, dst)
print(dst)
