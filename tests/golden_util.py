"""Helpers around tests/golden/*.npz (written by tools/make_golden.py from the compiled reference)."""
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CALL_KEYS = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr",
             "refmed", "refsd", "refiqr")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return g, json.loads(str(g["plan"])), json.loads(str(g["flags"]))


def calls_from_array(a):
    out = []
    for row in np.asarray(a).reshape(-1, len(CALL_KEYS)):
        d = {k: (int(v) if i < 7 else float(v)) for i, (k, v) in enumerate(zip(CALL_KEYS, row))}
        out.append(d)
    return out


def regenerate_inputs(lib, plan, g):
    """Re-create the case's inputs from the plan and check them against the recorded hashes."""
    from rsicnv_amd import synth
    plan = dict(plan)
    plan["events"] = [tuple(e) for e in plan["events"]]
    plan["nruns"] = [tuple(e) for e in plan["nruns"]]
    plan["lower"] = [tuple(e) for e in plan["lower"]]
    fasta, depth = synth.generate_host(lib, plan)
    assert sha(fasta) == str(g["fasta_sha"]), "synthetic FASTA differs from the one the golden file was made with"
    assert sha(depth) == str(g["depth_sha"]), "synthetic depth differs from the one the golden file was made with"
    return fasta, depth
