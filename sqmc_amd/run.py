"""python -m sqmc_amd.run -i <deck>   (or  < deck)

Runs a reference input deck on the GPU path.  run_type `none`: a projector Monte Carlo walk, see
walk_run.py.  run_type `hci`: the deck grammar of read_input
(do_walk.f90:406-574 for hci decks: seeds, run_type, `eps_var eps_pt target_error n_states`,
dump_wf_var, the chem Hamiltonian block of chemistry.f90:119-245, then the namelists
&selected_ci and &hf_det) and the result lines of perform_hci (hci.f90:323, 487, 833-841), so the
shipped decks (C2_v2z_curve/*/i_1sigma_g) run unchanged next to their FCIDUMP and the same
`grep 'Total energy(1)'` post-processing applies.  PT: deterministic, or with `n_mc > 0` and
`eps_pt_big` in &selected_ci the semistochastic scheme of second_order_pt_alias.  Host-side
plumbing: every piece that scales with the number of determinants runs in libsqmc_gpu."""
import argparse
import os
import re
import sys
import time

import numpy as np


def _strip(line):
    return line.split("!")[0] if line.lstrip().startswith("!") else line


def _logical(tok):
    t = tok.strip().strip(".").lower()
    if t[:1] in ("t", "f"):
        return t[0] == "t"
    raise ValueError("not a Fortran logical: " + tok)


def _numbers(line, n):
    """first n list-directed numeric items of a line (Fortran accepts 1.e-4, 1d-4, commas)"""
    out = []
    for tok in re.split(r"[,\s]+", line.strip()):
        try:
            out.append(float(tok.lower().replace("d", "e")))
        except ValueError:
            break
        if len(out) == n:
            break
    if len(out) < n:
        raise ValueError("expected %d numbers in %r" % (n, line))
    return out


def _namelists(text):
    """&name key=value ... /  ->  {name: {key: [values]}} with r*c repeat counts expanded"""
    groups = {}
    for m in re.finditer(r"^\s*&(\w+)(.*?)/", text, re.S | re.M):
        body, d = m.group(2), {}
        for km in re.finditer(r"(\w+)\s*=\s*([^=]*?)(?=\s+\w+\s*=|\s*$)", body.strip(), re.S):
            vals = []
            for tok in re.split(r"[,\s]+", km.group(2).strip()):
                if not tok:
                    continue
                rep, _, v = tok.partition("*") if "*" in tok else ("1", "", tok)
                vals += [v] * int(rep)
            d[km.group(1).lower()] = vals
        groups[m.group(1).lower()] = d
    return groups


def parse_hci_deck(text):
    lines = [l for l in text.splitlines() if l.strip() and not l.lstrip().startswith("!")]
    deck = {}
    # The decks of src/e2e_tests were written for the grammar in which the walk parameters (six lines:
    # nstep..., w_abs_gen..., tau..., reweight..., population control..., proposal_method...) come before
    # run_type, and semistoch/use_exp_proj after dump_wf_var: recognise it by the position of 'hci'.
    if lines[1].split()[0].strip("'\"").lower() != "hci" and len(lines) > 7 and lines[7].split()[0].strip("'\"").lower() == "hci":
        lines = [lines[0]] + lines[7:10] + lines[11:]
    s = re.sub(r"\s+", " ", lines[0].strip())
    digits = "".join(ch for ch in lines[0][:33] if ch.isdigit())           # '(4i4,x,4i4)'
    deck["irand_seed"] = [[int(digits[4 * i:4 * i + 4]) for i in range(4)], [int(digits[16 + 4 * i:20 + 4 * i]) for i in range(4)]]
    deck["run_type"] = lines[1].split()[0].strip("'\"").lower()
    if deck["run_type"] != "hci":
        raise SystemExit("sqmc_amd.run: only run_type hci decks are driven from a deck file (walks: sqmc_amd.host.GpuWalk / the C ABI)")
    ev, ep, te, ns = _numbers(lines[2], 4)
    deck.update(eps_var=ev, eps_pt=ep, target_error=te, n_states=int(ns))
    deck["dump_wf_var"] = _logical(lines[3].split()[0])
    toks = lines[4].replace(",", " ").split()
    deck["hamiltonian_type"], deck["ipr"] = toks[0].strip("'\"").lower(), int(float(toks[1]))
    nl = _namelists("\n".join(lines[5:]))
    sci = nl.get("selected_ci", {})
    deck["eps_var_sched"] = [float(v.lower().replace("d", "e")) for v in sci.get("eps_var_sched", [])]
    deck["n_mc"] = int(float(sci.get("n_mc", ["0"])[0]))
    deck["eps_pt_big"] = float(sci["eps_pt_big"][0].lower().replace("d", "e")) if "eps_pt_big" in sci else 0.0
    if deck["hamiltonian_type"] == "heg":              # read_heg, heg.f90:102-170
        deck["n_dim"] = int(_numbers(lines[5], 1)[0]); deck["r_s"] = _numbers(lines[6], 1)[0]
        ne, nu = _numbers(lines[7], 2)
        deck.update(nelec=int(ne), nup=int(nu), cutoff_radius=_numbers(lines[8], 1)[0])
        return deck
    if deck["hamiltonian_type"] != "chem":
        raise SystemExit("sqmc_amd.run: hamiltonian_type %r decks are not handled here" % deck["hamiltonian_type"])
    ne, nu = _numbers(lines[5], 2)
    deck.update(nelec=int(ne), nup=int(nu), point_group=lines[6].split()[0].strip("'\"").lower(), time_sym=_logical(lines[7].split()[0]))
    deck["z"] = int(_numbers(lines[8], 1)[0])
    deck["norb"] = int(_numbers(lines[9], 1)[0])
    deck["orbital_symmetries"] = [int(x) for x in _numbers(lines[10], deck["norb"])]
    hf = nl.get("hf_det", {})
    deck["hf_symmetry"] = int(hf["hf_symmetry"][0]) if "hf_symmetry" in hf else None
    if "irreps" in hf:
        raise SystemExit("sqmc_amd.run: &hf_det irreps=... (hand-picked occupations) is not handled; use hf_symmetry")
    return deck


def run_hci_heg(deck, out=sys.stdout):
    """HEG decks: no integral file; result lines as hci.f90 prints them for 'heg' (no state index in the
    older output format the e2e fixtures were produced with; Madelung total, correlation energy)."""
    import torch            # noqa: F401
    import sqmc_amd
    from . import host as H
    p = lambda *a: (print(*a, file=out), out.flush())
    sqmc_amd.set_device(0)
    h = H.HegHost(deck["n_dim"], deck["r_s"], deck["nelec"], deck["nup"], deck["cutoff_radius"])
    p("Within cutoff_radius =%10.5f number of spatial orbitals =%4d" % (deck["cutoff_radius"], h.norb))
    g = h.gpu()
    e_hf = g.hamiltonian_batch([h.hf_up], [h.hf_dn], [h.hf_up], [h.hf_dn])[0]
    mad = h.madelung_energy() if deck["n_dim"] == 3 else 0.0
    p("Madelung energy =%10.6f" % mad)
    p("Iteration   0 %8d =%9.2E dets, energy=%16.6f" % (1, 1.0, e_hf))

    def log(msg):
        m = re.match(r"Iteration\s+(\d+) eps1=(\S+) ndets=\s*(\d+).*energy=(.*)", msg)
        p("Iteration%4d %8d =%9.2E dets, energy=%s" % (int(m.group(1)), int(m.group(3)), float(m.group(3)), "".join("%16.6f" % float(x) for x in m.group(4).split())))
    up, dn, wts, energy, hist = H.hci_variational(h, g, deck["eps_var"], eps_sched=tuple(deck["eps_var_sched"]), n_states=deck["n_states"], log=log)
    res = {"ndets": len(up), "hist": hist, "e_hf": float(e_hf), "madelung": mad}
    e0 = float(energy[0])
    if deck["n_mc"] > 0 and deck["eps_pt_big"] > deck["eps_pt"]:
        r = H.hci_pt2_stochastic(h, g, up, dn, wts[:, 0], e0, deck["eps_pt"], deck["eps_pt_big"], deck["n_mc"], deck["target_error"],
                                 seed=deck["irand_seed"][0], log=lambda m: p("\n" + m))
        de, err = r["pt_big"] + r["pt_diff"], r["pt_diff_std_dev"]
        p("\nVariational energy=%s%15.9f" % (" " * 16, e0))
        p("Second-order PT energy lowering=%s%15.9f +-%12.9f (%13.9f%13.9f)" % (" " * 3, de, err, r["pt_big"], r["pt_diff"]))
        p("Total energy=%s%15.9f +-%12.9f" % (" " * 22, e0 + de, err))
        p("Total energy (includ. Madelung)=%s%15.9f +-%12.9f" % (" " * 3, e0 + de + mad, err))
        res.update(pt=de, pt_err=err, pt_big=r["pt_big"], pt_diff=r["pt_diff"], n_samples=len(r["samples"]))
    else:
        de, nconn = H.hci_pt2(h, g, up, dn, wts[:, 0], e0, deck["eps_pt"])
        p("\nPT_correction, eps_pt, ndets_connected for fully deterministic run=%15.9f%12.4E%12d" % (de, deck["eps_pt"], nconn))
        p("\nVariational energy=%s%15.9f" % (" " * 16, e0))
        p("Second-order PT energy lowering=%s%15.9f" % (" " * 3, de))
        p("Total energy=%s%15.9f" % (" " * 22, e0 + de))
        p("Total energy (includ. Madelung)=%s%15.9f" % (" " * 3, e0 + de + mad))
        res.update(pt=de, n_connected=nconn)
    p("Correlation energy =%s%15.9f" % (" " * 16, e0 + de - e_hf))
    g.close()
    res.update(e_var=e0, e_total=e0 + de)
    return res


def run_hci(deck, fcidump="FCIDUMP", out=sys.stdout):
    if deck["hamiltonian_type"] == "heg":
        return run_hci_heg(deck, out)
    import torch            # noqa: F401  one libamdhip64 per process
    import sqmc_amd
    from . import host as H
    p = lambda *a: (print(*a, file=out), out.flush())
    sqmc_amd.set_device(0)
    t0 = time.perf_counter()
    h = H.ChemHost(fcidump, deck["nelec"], deck["nup"], deck["point_group"], time_sym=deck["time_sym"], z=deck["z"], hf_symmetry=deck["hf_symmetry"])
    if h.norb != deck["norb"]:
        raise SystemExit("norb of the deck (%d) and of %s (%d) differ" % (deck["norb"], fcidump, h.norb))
    if list(h.orbsym_file[1:]) != deck["orbital_symmetries"]:
        raise SystemExit("orbital_symmetries of the deck and ORBSYM of %s differ" % fcidump)
    n_states, eps_var = deck["n_states"], deck["eps_var"]
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    wf = H.wf_filename(min([eps_var] + deck["eps_var_sched"]))
    if os.path.exists(wf):
        p("\nReading variational wavefn from " + wf)
        up, dn, wts, energy = H.read_wf_var(wf, n_states)
    else:
        e_hf = g.hamiltonian_batch([h.hf_up], [h.hf_dn], [h.hf_up], [h.hf_dn])[0]
        sched = deck["eps_var_sched"] or [eps_var]
        p("Iteration   0 eps1=%s ndets=%9d =%9.2E energy=%s" % (_es71(sched[0]), 1, 1.0, "".join("%16.6f" % v for v in [e_hf] + [0.0] * (n_states - 1))))

        def log(msg):                                   # host.hci_variational reports one line per iteration
            m = re.match(r"Iteration\s+(\d+) eps1=(\S+) ndets=\s*(\d+).*energy=(.*)", msg)
            it, eps, nd, en = int(m.group(1)), float(m.group(2)), int(m.group(3)), [float(x) for x in m.group(4).split()]
            p("Iteration%4d eps1=%s ndets=%9d =%9.2E energy=%s" % (it, _es71(eps), nd, float(nd), "".join("%16.6f" % v for v in en)))
        up, dn, wts, energy, hist = H.hci_variational(h, g, eps_var, eps_sched=tuple(deck["eps_var_sched"]), n_states=n_states, log=log)
        if deck["dump_wf_var"]:
            p("\nWriting variational wavefn to " + wf)
            H.write_wf_var(wf, up, dn, wts, energy)
    g.close()
    t1 = time.perf_counter()
    results = []
    for i in range(n_states):
        if deck["n_mc"] > 0 and deck["eps_pt_big"] > deck["eps_pt"]:
            # semistochastic PT (second_order_pt_alias): deterministic at eps_pt_big + sampled difference; in the determinant basis
            plain, du, dd, dc = h, up, dn, wts[:, i]
            if h.time_sym:
                import copy
                plain = copy.copy(h); plain.time_sym = False
                du, dd, dc = H.time_symmetrized_to_dets(up, dn, wts[:, i], h.z)
            gp = plain.gpu()
            gp.set_hb_tables(*plain.hb_tables(gp))
            try:
                r = H.hci_pt2_stochastic(plain, gp, du, dd, dc, float(energy[i]), deck["eps_pt"], deck["eps_pt_big"], deck["n_mc"], deck["target_error"],
                                         seed=deck["irand_seed"][0], log=lambda m: p("\n" + m))
            finally:
                gp.close()
            de, nconn = r["pt_big"] + r["pt_diff"], r["n_connected_big"]
            p("\nState%4d:" % (i + 1))
            p("Variational energy(%d)=%s%15.9f" % (i + 1, " " * 12, energy[i]))
            p("2nd-order PT energy lowering(%d)=%s%15.9f +-%12.9f (%13.9f%13.9f)" % (i + 1, " " * 2, de, r["pt_diff_std_dev"], r["pt_big"], r["pt_diff"]))
            p("Total energy(%d)=%s%15.9f +-%12.9f" % (i + 1, " " * 18, energy[i] + de, r["pt_diff_std_dev"]))
            results.append((float(energy[i]), float(de), int(nconn)))
            continue
        de, nconn = H.hci_pt2_determinant_basis(h, up, dn, wts[:, i], float(energy[i]), deck["eps_pt"])
        p("\nState%4d:" % (i + 1))
        p("Variational energy(%d)=%s%15.9f" % (i + 1, " " * 12, energy[i]))
        p("2nd-order PT energy lowering(%d)=%s%15.9f" % (i + 1, " " * 2, de))
        p("Total energy(%d)=%s%15.9f" % (i + 1, " " * 18, energy[i] + de))
        p("eps_var, eps_pt, ndets, ndets_connected(total), Variational, PT_actv, PT_full, Total Energies(%d)=%8.1E%8.1E%9d%11d%16.9f%15.9f%15.9f%16.9f"
          % (i + 1, eps_var, deck["eps_pt"], len(up), nconn, energy[i], de, de, energy[i] + de))
        results.append((float(energy[i]), float(de), int(nconn)))
    p("\nsqmc_amd: variational stage %.2f s, PT stage %.2f s on the GPU path" % (t1 - t0, time.perf_counter() - t1))
    return dict(ndets=len(up), states=results)


def _es71(x):
    """Fortran es7.1e1"""
    m, e = ("%.1e" % x).split("e")
    return "%sE%s%d" % (m, "-" if int(e) < 0 else "+", abs(int(e)))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m sqmc_amd.run", description=__doc__.split("\n\n")[0])
    ap.add_argument("-i", "--input", default=None, help="deck file (default: stdin)")
    ap.add_argument("--fcidump", default="FCIDUMP", help="integral file (the reference reads ./FCIDUMP)")
    ap.add_argument("--walkalize", default=None, help="walk decks: write the per-step record of unit 1 (step, 1/rew_fac_inv, w_abs_gen, e_gen, nwalk) to this file")
    for name, what in (("psit-con", "C(T): Psi_T connections with local-energy pieces (psit_con_in/out_file, semistoch.f90:86-126)"),
                       ("dtm-elems", "the deterministic space and its Hamiltonian (dtm_elems_in/out_file, do_walk.f90:898-1010)")):
        ap.add_argument("--%s-in" % name, default=None, help="walk decks: read %s from this file instead of building it" % what)
        ap.add_argument("--%s-out" % name, default=None, help="walk decks: write %s to this file" % what)
    a = ap.parse_args(argv)
    text = open(a.input).read() if a.input else sys.stdin.read()
    lines = [l for l in text.splitlines() if l.strip() and not l.lstrip().startswith("!")]
    run_type = lines[1].split()[0].strip("'\"").lower() if len(lines) > 1 else ""
    if run_type in ("none", "no_fixed_node"):          # a projector walk: the grammar of read_input for run_type /= hci
        from .walk_run import parse_walk_deck, run_walk
        return run_walk(parse_walk_deck(text), a.fcidump, walkalize=a.walkalize, psit_con_in=a.psit_con_in, psit_con_out=a.psit_con_out,
                        dtm_elems_in=a.dtm_elems_in, dtm_elems_out=a.dtm_elems_out)
    deck = parse_hci_deck(text)
    return run_hci(deck, a.fcidump)


if __name__ == "__main__":
    main()
