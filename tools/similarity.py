#!/usr/bin/env python3
"""Token-level similarity of a Python file of this repo with a reference file (comments and docstrings stripped):
difflib ratio over the token sequences and the share of this file's tokens that sit in identical runs of >= 12 tokens.
Only runs in the build container (the reference tree is not shipped):  python tools/similarity.py karios_amd/matcher/klt.py /root/reference/karios/matcher/klt.py"""
import difflib
import io
import sys
import tokenize


def tokens(path):
    out, prev = [], None
    for tok in tokenize.generate_tokens(io.StringIO(open(path).read()).readline):
        if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER):
            prev = tok.type if tok.type != tokenize.COMMENT else prev
            continue
        if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.NL):
            prev = tok.type
            continue   # docstring / bare string statement
        out.append(tok.string)
        prev = tok.type
    return out


def main():
    a, b = tokens(sys.argv[1]), tokens(sys.argv[2])
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    long_runs = sum(m.size for m in sm.get_matching_blocks() if m.size >= 12)
    print(f"{sys.argv[1]}: {len(a)} tokens, reference {len(b)} tokens, ratio {sm.ratio():.3f}, "
          f"{100.0 * long_runs / max(1, len(a)):.1f} % of tokens in identical runs >= 12")


if __name__ == "__main__":
    main()
