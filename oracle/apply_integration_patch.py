#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — applies INTEGRATION.md's patch to COPIES of three reference files, into oracle/_ref/patched/ (git-ignored
build output; the reference tree is never written and no reference source is committed):

  src/core/models/haplotype_likelihood_array.cpp   the two populate() definitions (:51-103, :105-199) are cut out and replaced by
                                                   #include "oracle/integration/populate_on_device.inc" (pack -> oct_phmm_populate -> scatter)
  src/core/models/haplotype_likelihood_model.hpp   + `friend class HaplotypeLikelihoodArray;` (the six penalty vectors reset() prepares)
  src/core/models/haplotype_likelihood_array.hpp   unchanged copy (so that its `#include "haplotype_likelihood_model.hpp"` finds the line above)

    python oracle/apply_integration_patch.py [/root/reference] [oracle/_ref/patched]
"""
import re
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
out = Path(sys.argv[2] if len(sys.argv) > 2 else HERE / "_ref" / "patched")
src = ref / "src" / "core" / "models"
dst = out / "core" / "models"
dst.mkdir(parents=True, exist_ok=True)


def function_span(text: str, signature_start: str, begin: int = 0):
    """[start, end) of the definition that begins with `signature_start`: up to the brace that closes its body."""
    start = text.index(signature_start, begin)
    i = text.index("{", text.index(")", start))
    depth = 0
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return start, i + 1
        i += 1


cpp = (src / "haplotype_likelihood_array.cpp").read_text()
a0, a1 = function_span(cpp, "void HaplotypeLikelihoodArray::populate(const ReadMap& reads")
b0, b1 = function_span(cpp, "void HaplotypeLikelihoodArray::populate(const TemplateMap& reads", a1)
assert a0 < a1 <= b0 < b1 and cpp[a1:b0].strip() == "", "the two populate definitions are expected to be adjacent"
inc = (HERE / "integration" / "populate_on_device.inc").resolve()
# the patch file carries its own `namespace octopus { ... }`: close the file's namespace around it
patched = (cpp[:a0] + "} // namespace octopus\n\n#include \"" + str(inc) + "\"\n\nnamespace octopus {\n" + cpp[b1:])
(dst / "haplotype_likelihood_array.cpp").write_text(patched)

hpp = (src / "haplotype_likelihood_model.hpp").read_text()
m = re.search(r"class HaplotypeLikelihoodModel\s*\{\s*public:", hpp)
assert m, "class HaplotypeLikelihoodModel { public: not found"
hpp = hpp[:m.start()] + "class HaplotypeLikelihoodArray;\n\n" + hpp[m.start():m.end()] + \
    "\n    friend class HaplotypeLikelihoodArray;   // INTEGRATION patch: populate() hands the six penalty vectors to the device\n" + hpp[m.end():]
(dst / "haplotype_likelihood_model.hpp").write_text(hpp)
(dst / "haplotype_likelihood_array.hpp").write_text((src / "haplotype_likelihood_array.hpp").read_text())
print(f"patched copies in {dst}: populate() bodies {a1 - a0} + {b1 - b0} characters -> {inc.name}")
