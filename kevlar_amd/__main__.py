"""`python -m kevlar_amd <cmd> ...`: the dispatcher is kevlar_amd.cli.run."""
from kevlar_amd.cli import run as main

if __name__ == '__main__':
    main()
