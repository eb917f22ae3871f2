"""Token-level similarity of kevlar_amd/*.py against the same-named reference files (comments and docstrings
stripped, kevlar_amd -> kevlar): the check the round-1 verdict applied.  Runs in the build container only
(/root/reference is not on the GPU box).  python scratch/toksim.py [files...]"""
import difflib, io, os, sys, tokenize

REF = '/root/reference/kevlar'
OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'kevlar_amd')


def tokens(path):
    src = open(path).read().replace('kevlar_amd', 'kevlar')
    out, prev = [], None
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER):
            prev = tok.type if tok.type in (tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT) else prev
            continue
        if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT) and tok.string[:3] in ('"""', "'''"):
            continue        # docstring
        out.append(tok.string)
        prev = tok.type
    return out


names = sys.argv[1:] or sorted(f for f in os.listdir(OURS) if f.endswith('.py'))
for name in names:
    ref = os.path.join(REF, name)
    if name == 'sequence.py':
        ref = os.path.join(REF, 'sequence.pyx')
    if not os.path.exists(ref):
        continue
    try:
        a, b = tokens(os.path.join(OURS, name)), tokens(ref)
    except (tokenize.TokenError, SyntaxError, IndentationError):
        continue
    ratio = difflib.SequenceMatcher(None, a, b, autojunk=False).ratio()
    inside = sum(blk.size for blk in difflib.SequenceMatcher(None, a, b, autojunk=False).get_matching_blocks()) / max(1, len(a))
    print('{:16s} ratio {:.2f}   share of our tokens matched in the reference {:.2f}   ({} / {} tokens)'.format(name, ratio, inside, len(a), len(b)))
