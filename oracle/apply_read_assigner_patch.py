#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — the second seam of INTEGRATION.md (section 3): src/core/tools/read_assigner.cpp:145-287, the functions that compute the
ploidy x reads likelihood matrix of read assignment. Writes into oracle/_ref/patched/ (git-ignored build output; the reference tree is never written and no
reference source is committed):

  read_assigner_seam_ref.inc      the reference's own functions estimate_max_indel_size* ... calculate_likelihoods(genotype, reads, model, workers), cut out of
                                  a copy of the file as they are
  read_assigner_seam_patched.inc  the same helpers, with the LAST function (:251-287) replaced by #include "oracle/integration/read_assigner_on_device.inc"
                                  (expand -> reset -> pack -> ONE oct_phmm_populate)
  core/models/haplotype_likelihood_model.hpp   gets `friend struct octopus::ReadAssignerDevice;` (the struct lives in read_assigner.cpp's unnamed namespace) next to the line apply_integration_patch.py adds

oracle/ref_assigner_bridge.cpp includes one or the other between stand-in types; tests/test_integration_patch.py compares the two.

    python oracle/apply_read_assigner_patch.py [/root/reference] [oracle/_ref/patched]
"""
import re
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
out = Path(sys.argv[2] if len(sys.argv) > 2 else HERE / "_ref" / "patched")
out.mkdir(parents=True, exist_ok=True)


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


cpp = (ref / "src" / "core" / "tools" / "read_assigner.cpp").read_text()
first = cpp.index("template <typename MappableTp>\nGenomicRegion::Size estimate_max_indel_size_helper")
sig = "template <typename Container>\nauto calculate_likelihoods(const Genotype<Haplotype>& genotype,"
last0, last1 = function_span(cpp, sig, first)
assert first < last0 < last1
(out / "read_assigner_seam_ref.inc").write_text(cpp[first:last1] + "\n")
inc = (HERE / "integration" / "read_assigner_on_device.inc").resolve()
(out / "read_assigner_seam_patched.inc").write_text(cpp[first:last0] + '#include "' + str(inc) + '"\n')

# the model's header: one more friend (the copy apply_integration_patch.py made, or a fresh one)
dst = out / "core" / "models"
dst.mkdir(parents=True, exist_ok=True)
hpp_path = dst / "haplotype_likelihood_model.hpp"
hpp = hpp_path.read_text() if hpp_path.exists() else (ref / "src" / "core" / "models" / "haplotype_likelihood_model.hpp").read_text()
if "friend struct octopus::ReadAssignerDevice;" not in hpp:
    m = re.search(r"class HaplotypeLikelihoodModel\s*\{\s*public:", hpp)
    assert m, "class HaplotypeLikelihoodModel { public: not found"
    hpp = hpp[:m.start()] + "namespace { struct ReadAssignerDevice; }\n\n" + hpp[m.start():m.end()] + \
        "\n    friend struct octopus::ReadAssignerDevice;   // INTEGRATION patch, second seam: read_assigner.cpp hands the six penalty vectors to the device\n" + hpp[m.end():]
    hpp_path.write_text(hpp)
print(f"seam of read_assigner.cpp: {last1 - first} characters, of which the last function's {last1 - last0} -> {inc.name}")
