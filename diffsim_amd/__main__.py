"""``python -m diffsim_amd`` -- the reference's cute_main.py / night_main.py loops on the MI355X engine (cli.py)."""
import sys

from .cli import main

if __name__ == "__main__":
    sys.exit(main())
