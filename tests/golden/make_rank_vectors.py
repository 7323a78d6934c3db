"""Transcribes the expected vectors of the reference's rank tests into tests/golden/rank_vectors.json.

Run in the build container (the reference checkout is not on the GPU box):
    python tests/golden/make_rank_vectors.py
Source: /root/reference/cpp/tests/sort/rank_test.cpp:62-430 (fixture Rank<T>: col1 = {5,4,3,5,8,5}, col2 = the same
with row 2 null; one TYPED_TEST per (method, order/null_policy/null_order, percentage) with the expected ranks of
col1 and col2 as literals).  Only parsing happens here: no value is computed."""
import json
import os
import re

SRC = "/root/reference/cpp/tests/sort/rank_test.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_vectors.json")
ARGS = {  # rank_test.cpp:47-56
    "asc_keep": dict(descending=False, null_include=False, null_before=False),
    "asc_top": dict(descending=False, null_include=True, null_before=True),
    "asc_bottom": dict(descending=False, null_include=True, null_before=False),
    "desc_keep": dict(descending=True, null_include=False, null_before=True),
    "desc_top": dict(descending=True, null_include=True, null_before=False),
    "desc_bottom": dict(descending=True, null_include=True, null_before=True),
}
METHODS = {"FIRST": 0, "AVERAGE": 1, "MIN": 2, "MAX": 3, "DENSE": 4}  # cudf::rank_method


def num(tok):
    tok = tok.strip()
    if "/" in tok:
        a, b = tok.split("/")
        return float(a) / float(b)
    return float(tok)


def parse_wrapper(body, name):
    m = re.search(name + r"\s*\{(.*?)\};", body, re.S)
    txt = m.group(1)
    lists = re.findall(r"\{([^{}]*)\}", txt)
    vals = [num(t) for t in lists[0].split(",")]
    valid = [t.strip() in ("true", "1") for t in lists[1].split(",")] if len(lists) > 1 else [True] * len(vals)
    return vals, valid


def main():
    lines = open(SRC).read().split("\n")
    text = "\n".join(lines)
    cases = []
    for m in re.finditer(r"TYPED_TEST\(Rank, (\w+)\)\n\{(.*?)\n\}\n", text, re.S):
        name, body = m.group(1), m.group(2)
        call = re.search(r"run_all_tests\(cudf::rank_method::(\w+),\s*(\w+),[^;]*?(,\s*true)?\);", body, re.S)
        line = text[: m.start()].count("\n") + 1
        c1, v1 = parse_wrapper(body, "col1_rank")
        c2, v2 = parse_wrapper(body, "col2_rank")
        cases.append(dict(name=name, line=line, method=METHODS[call.group(1)], percentage=bool(call.group(3)),
                          **ARGS[call.group(2)], col1=c1, col1_valid=v1, col2=c2, col2_valid=v2))
    json.dump({"source": "cpp/tests/sort/rank_test.cpp", "input": [5, 4, 3, 5, 8, 5], "col2_valid": [1, 1, 0, 1, 1, 1],
               "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)


if __name__ == "__main__":
    main()
