#!/usr/bin/env python
"""Real-dictionary parity in one command (GPU box).

Token-level parity against the reference binary is unpinned in this repository's own tests: neither the reference
(Rust) nor its IPADIC dictionary can be obtained here.  The day both exist somewhere, run there

    kanpyo tokenize < sentences.txt > expected.txt          # the reference CLI (src/bin/kanpyo.rs:106-126,174-197)
    python tools/real_ipadic_check.py ipa.dict sentences.txt expected.txt

and this script tokenizes the same lines on the GPU from the same Kanpyo `.dict` (kanpyo_amd.dictfile.load_dict reads
the reference's zip container), prints them in the CLI's `surface\\tfeat,feat,...` format and diffs the two streams
line by line.  Exit status 0 = identical.  Each input line is stripped of trailing whitespace as the CLI does
(src/bin/kanpyo.rs:122).
"""
import argparse
import difflib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dict", help="Kanpyo .dict file (zip of the six blobs, kanpyo-dict/src/dict.rs:51-69)")
    ap.add_argument("sentences", help="UTF-8 text, one sentence per line")
    ap.add_argument("expected", help="output of `kanpyo tokenize` fed with the same file")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--show", type=int, default=40, help="diff lines to print")
    args = ap.parse_args()

    try:
        import torch  # noqa: F401  (its HIP runtime first)
    except Exception:
        pass
    from kanpyo_amd import Tokenizer
    from kanpyo_amd.dictfile import format_tokens, load_dict

    df = load_dict(args.dict)
    tok = Tokenizer(df.dict, device=args.device)
    with open(args.sentences, encoding="utf-8") as f:
        lines = [ln.rstrip() for ln in f.read().split("\n")]
    if lines and lines[-1] == "":
        lines.pop()  # the CLI stops at EOF, not at a final empty read
    got = []
    for lo in range(0, len(lines), args.batch):
        for toks in tok.tokenize_batch(lines[lo : lo + args.batch]):
            got += format_tokens(toks, df).split("\n") if toks else []
    with open(args.expected, encoding="utf-8") as f:
        exp = f.read().split("\n")
    if exp and exp[-1] == "":
        exp.pop()
    if got == exp:
        print(f"identical: {len(lines)} sentences, {len(got)} output lines")
        return 0
    diff = list(difflib.unified_diff(exp, got, "kanpyo (reference)", "kanpyo_amd (GPU)", lineterm="", n=2))
    print("\n".join(diff[: args.show]))
    print(f"... DIFFERENT: {len(exp)} expected lines, {len(got)} produced, {sum(1 for d in diff if d[:1] in '+-' and d[:3] not in ('+++', '---'))} changed")
    return 1


if __name__ == "__main__":
    sys.exit(main())
