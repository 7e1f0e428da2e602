#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — the ONE owner of oracle/_ref/patched/ (git-ignored build output; the reference tree is never written and no
reference source is committed). Every file below is regenerated from the REFERENCE's sources on every run — never from an earlier output —
so the result does not depend on what was there before or on the order of anything; `.stamp` is written last and is what oracle/Makefile's
rules depend on.

First seam, HaplotypeLikelihoodArray::populate:
  core/models/haplotype_likelihood_array.cpp   src/core/models/haplotype_likelihood_array.cpp with the two populate() definitions (:51-103, :105-199)
                                               RENAMED populate_on_host (the fallback for regions the device path does not take) and followed by
                                               #include "integration/populate_on_device.inc": the new populate() bodies (pack -> oct_phmm_populate -> scatter)
  core/models/haplotype_likelihood_array.hpp   copy + the two populate_on_host declarations (and so that its `#include "haplotype_likelihood_model.hpp"` finds the header below)
Second seam, src/core/tools/read_assigner.cpp:145-287:
  read_assigner_seam_ref.inc                   the reference's own functions estimate_max_indel_size* ... calculate_likelihoods(genotype, reads, model, workers),
                                               cut out of a copy of the file as they are
  read_assigner_seam_patched.inc               the same helpers, with the LAST function (:251-287) renamed calculate_likelihoods_on_host (the fallback) and followed by
                                               #include "integration/read_assigner_on_device.inc" (the new last function: expand -> reset -> pack -> ONE oct_phmm_populate)
Third seam, src/core/tools/read_realigner.cpp:83-155:
  read_realigner_seam_ref.inc                  the reference's own compute_read_hashes, the two realign(read, haplotype, ...) helpers and
                                               realign(reads, haplotype, model, log_likelihoods, workers), cut out of a copy of the file as they are
  read_realigner_seam_patched.inc              the same helpers, with the LAST function (:114-155) replaced by
                                               #include "integration/read_realigner_on_device.inc" (reset -> pack -> ONE oct_phmm_align -> AlignedRead::realign)
All seams:
  core/models/haplotype_likelihood_model.hpp   src/core/models/haplotype_likelihood_model.hpp + one `friend` line per seam (FRIENDS below): the seams hand the six
                                               penalty vectors reset() prepares to the device

    python oracle/make_patched_tree.py [/root/reference] [oracle/_ref/patched]
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
stamp = out / ".stamp"
if stamp.exists():
    stamp.unlink()                                  # a run that dies half-way leaves no stamp

# (forward declaration in namespace octopus, friend line in class HaplotypeLikelihoodModel) per seam
FRIENDS = [
    ("class HaplotypeLikelihoodArray;",
     "friend class HaplotypeLikelihoodArray;   // INTEGRATION patch: populate() hands the six penalty vectors to the device"),
    ("namespace { struct ReadAssignerDevice; }",   # the struct lives in read_assigner.cpp's unnamed namespace
     "friend struct octopus::ReadAssignerDevice;   // INTEGRATION patch, second seam: read_assigner.cpp hands the six penalty vectors to the device"),
    ("namespace { struct ReadRealignerDevice; }",   # ... and this one in read_realigner.cpp's
     "friend struct octopus::ReadRealignerDevice;   // INTEGRATION patch, third seam: read_realigner.cpp hands the six penalty vectors to the device"),
]


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


# ---- first seam
cpp = (src / "haplotype_likelihood_array.cpp").read_text()
a0, a1 = function_span(cpp, "void HaplotypeLikelihoodArray::populate(const ReadMap& reads")
b0, b1 = function_span(cpp, "void HaplotypeLikelihoodArray::populate(const TemplateMap& reads", a1)
assert a0 < a1 <= b0 < b1 and cpp[a1:b0].strip() == "", "the two populate definitions are expected to be adjacent"
inc = (HERE.parent / "integration" / "populate_on_device.inc").resolve()
# The reference's own two bodies stay in the file under another name, populate_on_host: what the patched populate() falls back to for a region the device path
# does not take (OCT_PHMM_EUNSUPPORTED: a read of 32 k bases, a 40 k-base haplotype) - the patched class must answer every call the unpatched one answers.
# The patch file carries its own `namespace octopus { ... }`: close the file's namespace around it.
HOST = "void HaplotypeLikelihoodArray::populate_on_host("
own_bodies = cpp[a0:b1].replace("void HaplotypeLikelihoodArray::populate(", HOST)
assert own_bodies.count(HOST) == 2
patched = (cpp[:a0] + own_bodies + "\n\n} // namespace octopus\n\n#include \"" + str(inc) + "\"\n\nnamespace octopus {\n" + cpp[b1:])
(dst / "haplotype_likelihood_array.cpp").write_text(patched)
# ... and the header declares them (the two populate declarations once more, renamed, without their default arguments)
ahpp = (src / "haplotype_likelihood_array.hpp").read_text()
decls = list(re.finditer(r"\n    void populate\((const (?:ReadMap|TemplateMap)& reads,[^;]*?)\);", ahpp))
assert len(decls) == 2, "the two populate declarations of haplotype_likelihood_array.hpp"
extra = "".join("\n    void populate_on_host(" + re.sub(r"\s*=\s*boost::none", "", d.group(1)) + ");   // INTEGRATION patch: the reference's own body, the fallback" for d in decls)
ahpp = ahpp[:decls[1].end()] + extra + ahpp[decls[1].end():]
(dst / "haplotype_likelihood_array.hpp").write_text(ahpp)

# ---- second seam
asg = (ref / "src" / "core" / "tools" / "read_assigner.cpp").read_text()
first = asg.index("template <typename MappableTp>\nGenomicRegion::Size estimate_max_indel_size_helper")
sig = "template <typename Container>\nauto calculate_likelihoods(const Genotype<Haplotype>& genotype,"
last0, last1 = function_span(asg, sig, first)
assert first < last0 < last1
(out / "read_assigner_seam_ref.inc").write_text(asg[first:last1] + "\n")
asg_inc = (HERE.parent / "integration" / "read_assigner_on_device.inc").resolve()
# (the reference's own last function stays, renamed calculate_likelihoods_on_host: the fallback for a genotype / read set the device path refuses, OCT_PHMM_EUNSUPPORTED)
own_last = asg[last0:last1].replace("auto calculate_likelihoods(const Genotype<Haplotype>& genotype,", "auto calculate_likelihoods_on_host(const Genotype<Haplotype>& genotype,")
assert own_last != asg[last0:last1]
(out / "read_assigner_seam_patched.inc").write_text(asg[first:last0] + own_last + '\n\n#include "' + str(asg_inc) + '"\n')

# ---- third seam: src/core/tools/read_realigner.cpp:83-155
rea = (ref / "src" / "core" / "tools" / "read_realigner.cpp").read_text()
r_first = rea.index("auto compute_read_hashes(const std::vector<AlignedRead>& reads")
r_sig = ("void realign(std::vector<AlignedRead>& reads, const Haplotype& haplotype,\n             HaplotypeLikelihoodModel model,\n"
         "             boost::optional<std::vector<HaplotypeLikelihoodModel::LogProbability>&> log_likelihoods,")
r_last0, r_last1 = function_span(rea, r_sig, r_first)
assert r_first < r_last0 < r_last1
(out / "read_realigner_seam_ref.inc").write_text(rea[r_first:r_last1] + "\n")
rea_inc = (HERE.parent / "integration" / "read_realigner_on_device.inc").resolve()
(out / "read_realigner_seam_patched.inc").write_text(rea[r_first:r_last0] + '#include "' + str(rea_inc) + '"\n')

# ---- the model's header, with every seam's friend line
hpp = (src / "haplotype_likelihood_model.hpp").read_text()
m = re.search(r"class HaplotypeLikelihoodModel\s*\{\s*public:", hpp)
assert m, "class HaplotypeLikelihoodModel { public: not found"
hpp = (hpp[:m.start()] + "".join(fwd + "\n" for fwd, _ in FRIENDS) + "\n" + hpp[m.start():m.end()] + "\n" +
       "".join("    " + line + "\n" for _, line in FRIENDS) + hpp[m.end():])
for _, line in FRIENDS:
    assert line in hpp
(dst / "haplotype_likelihood_model.hpp").write_text(hpp)

stamp.write_text("written by oracle/make_patched_tree.py\n")
print(f"patched copies in {out}: populate() bodies {a1 - a0} + {b1 - b0} characters -> {inc.name}; "
      f"read_assigner.cpp seam {last1 - first} characters, of which the last function's {last1 - last0} -> {asg_inc.name}; "
      f"read_realigner.cpp seam {r_last1 - r_first} characters, of which the last function's {r_last1 - r_last0} -> {rea_inc.name}")
