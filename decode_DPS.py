"""`python decode_DPS.py --task dna --sample_M 10` — same entry point name as the reference's decode_DPS.py;
the implementation is svdd_amd/cli.py (method "dps")."""
from svdd_amd.cli import main

if __name__ == "__main__":
    main("dps")
