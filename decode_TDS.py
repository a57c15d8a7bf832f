"""`python decode_TDS.py --task dna --sample_M 10` — same entry point name as the reference's decode_TDS.py;
the implementation is svdd_amd/cli.py (method "tds")."""
from svdd_amd.cli import main

if __name__ == "__main__":
    main("tds")
