"""`python -m kevlar_amd <cmd> ...` (the reference's kevlar/__main__.py:14-30)."""
import kevlar_amd


def main(arglist=None):
    args = kevlar_amd.cli.parse_args(arglist)
    if args.cmd is None:
        kevlar_amd.cli.parser().parse_args(['-h'])
    assert args.cmd in kevlar_amd.cli.mains
    kevlar_amd.plog('[kevlar] running version {}'.format(kevlar_amd.__version__))
    kevlar_amd.cli.mains[args.cmd](args)


if __name__ == '__main__':
    main()
