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


def check_hip_against_golden(hot, hotlib, name):
    """One case of tests/golden through the HIP path (hot: api.RsiHot), every array, scalar, status vector, call table and output
    row against what the compiled reference produced.  Returns the Result (for assertions about the path taken)."""
    from conftest import calls_equal
    from rsicnv_amd import api
    g, plan, flags = load(name)
    fasta, depth = regenerate_inputs(hotlib, plan, g)
    res = hot.run(api.make_params(**flags), depth, fasta)
    assert np.array_equal(res.noncode, g["noncode"])
    if flags.get("gcadjust", 1):
        assert sha(hot.fetch("rd_gc")) == str(g["rd_gc_sha"])
    assert sha(hot.fetch("rd_concat")) == str(g["rd_concat_sha"])
    assert (res.stats["RDmedian"], res.stats["RDsd"]) == tuple(g["chrom"])
    assert np.array_equal(hot.fetch("binmedint"), g["binmedint"])
    np.testing.assert_allclose(hot.fetch("binnb"), g["binnb"], rtol=1e-6, atol=0)
    trans = flags.get("trans", 0)
    pre = "med" if trans == 1 else "nb"      # the last scan run leaves its arrays on the device
    sc = g[f"{pre}_scan"]
    got = [res.stats[k] for k in ("tmedian1", "tsigma1", "tlamda1", "tmedian2", "tsigma2", "tlamda2")]
    np.testing.assert_allclose(got, sc[:6], rtol=1e-12)
    assert res.stats["Lmax"] == int(sc[7])
    for w in ("status1", "status1f", "status2"):
        assert np.array_equal(hot.fetch(w), g[f"{pre}_{w}"]), w
    for which in ("calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), calls_from_array(g[which]))
        assert ok, f"{which}: {why}"
    # the output rows, byte for byte as cnv_format1 prints them (rsi.cpp:581-631)
    assert "\n".join(res.format_rows("chrS")) + ("\n" if res.rows else "") == str(g["rows"])
    return res
