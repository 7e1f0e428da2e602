#!/usr/bin/env python3
"""Transcribe the reference's own known-answer vectors into a JSON fixture.

Source: /root/reference/test/unit/core/models/pair_hmm_tests.cpp (Boost.Test; not buildable here:
no Boost). Each `test = {...}; expected_alignment = {...}; CHECK_ALIGNER(test, hmm, ...)` block
becomes one record, tagged with the instantiations the reference checks it against
(SSE2PairHMM<B,short|int> :203-592, AVX2PairHMM<16,short> :596-712, speed cases :104-199,719-757).
All vectors use the no-SNV-mask overload with vector gap_open, scalar gap_extend, nuc_prior.

Run in the build container (the reference tree is not present on the GPU box):
    python tests/golden/make_pair_hmm_golden.py
"""
import json
import re
import sys
from pathlib import Path

SRC = Path("/root/reference/test/unit/core/models/pair_hmm_tests.cpp")
OUT = Path(__file__).with_name("pair_hmm_tests.json")


def ints(s):
    return [int(x) for x in re.findall(r"-?\d+", s)]


def parse_testcase(body):
    strs = re.findall(r'"([^"]*)"', body)
    lists = re.findall(r"\{([^{}]*)\}", body)
    tail = ints(re.sub(r'"[^"]*"', "", re.sub(r"\{[^{}]*\}", "", body)))
    assert len(strs) == 2 and len(lists) == 2 and len(tail) == 2, (strs, tail)
    return dict(truth=strs[0], target=strs[1], base_qualities=ints(lists[0]), gap_open=ints(lists[1]),
                gap_extend=tail[0], nuc_prior=tail[1])


def parse_alignment(body):
    strs = re.findall(r'"([^"]*)"', body)
    nums = ints(re.sub(r'"[^"]*"', "", body))
    assert len(strs) == 2 and len(nums) == 2
    return dict(score=nums[0], begin=nums[1], align1=strs[0], align2=strs[1])


def braces(text, start):
    """text[start] == '{' -> index one past the matching '}'"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def main():
    text = SRC.read_text()
    records = []
    # global speed cases
    glob = {}
    for m in re.finditer(r"^(TestCase|Alignment)\s+(\w+)\s*=\s*\{", text, re.M):
        end = braces(text, m.end() - 1)
        body = text[m.end():end - 1]
        glob[m.group(2)] = parse_testcase(body) if m.group(1) == "TestCase" else parse_alignment(body)
    # per test-case blocks
    for case in re.finditer(r"BOOST_AUTO_TEST_CASE\((\w+)\)\s*\{", text):
        end = braces(text, case.end() - 1)
        block = text[case.end():end]
        line0 = text.count("\n", 0, case.start()) + 1
        hmms = {}
        for m in re.finditer(r"(SSE2|AVX2|AVX512)PairHMM<\s*(\d+)\s*,\s*(short|int)\s*>\s+(\w+)\s*;", block):
            hmms[m.group(4)] = dict(isa=m.group(1).lower(), band=int(m.group(2)),
                                    score_bits=16 if m.group(3) == "short" else 32)
        cur_test = cur_exp = None
        pos = 0
        token = re.compile(r"(test|expected_alignment)\s*=\s*\{|CHECK_(ALIGNER|SPEED)\((\w+),\s*(\w+),\s*(\w+)")
        idx = 0
        while True:
            m = token.search(block, pos)
            if not m:
                break
            if m.group(1):
                e = braces(block, m.end() - 1)
                body = block[m.end():e - 1]
                if m.group(1) == "test":
                    cur_test = parse_testcase(body)
                    idx += 1
                else:
                    cur_exp = parse_alignment(body)
                pos = e
            else:
                tname, hname, ename = m.group(3), m.group(4), m.group(5)
                t = cur_test if tname == "test" else glob[tname]
                e = cur_exp if ename == "expected_alignment" else glob[ename]
                name = f"{case.group(1)}/{idx if tname == 'test' else tname}"
                rec = next((r for r in records if r["name"] == name and r["test"] == t), None)
                if rec is None:
                    rec = dict(name=name, source=f"pair_hmm_tests.cpp:{line0 + block.count(chr(10), 0, m.start())}",
                               test=t, expected=e, instantiations=[])
                    records.append(rec)
                if hmms[hname] not in rec["instantiations"]:
                    rec["instantiations"].append(hmms[hname])
                pos = m.end()
    for r in records:
        t = r["test"]
        b = r["instantiations"][0]["band"]
        assert len(t["truth"]) == len(t["target"]) + 2 * b - 1, r["name"]
        assert len(t["target"]) == len(t["base_qualities"]) and len(t["truth"]) == len(t["gap_open"]), r["name"]
    OUT.write_text(json.dumps(dict(source=str(SRC), records=records), indent=1) + "\n")
    print(f"wrote {len(records)} records to {OUT}", file=sys.stderr)


if __name__ == "__main__":
    main()
