"""`python decode_tweedie.py --task dna --sample_M 10` — same entry point name as the reference's decode_tweedie.py;
the implementation is svdd_amd/cli.py (method "tweedie")."""
from svdd_amd.cli import main

if __name__ == "__main__":
    main("tweedie")
