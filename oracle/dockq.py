"""Oracle: DockQ scoring of docked candidates (numpy, float64).  Test infrastructure only.

Restates what the reference's runner computes for every saved candidate
(/root/reference/AbDock/src/tools/runner/design_for_pdb.py:316-321 -> calc_DockQ(model, native, use_CA_only=True)):

  /root/reference/AbDock/DockQ/DockQ.py:98-385   calc_DockQ       (fnat, interface, iRMS, LRMS, DockQ)
  /root/reference/AbDock/DockQ/DockQ.py:18-49    parse_fnat       (interface = residues named on the NATIVE: lines)
  /root/reference/AbDock/DockQ/src/fnat.c:100-252                 (residue-residue contacts, Fnat / Fnonnat)
  /root/reference/AbDock/DockQ/src/molecule.c:581-612  crd()      (minimum heavy-atom distance of two residues, squared)

on tensors instead of PDB files: a structure is (pos [L, A, 3], mask [L, A] bool, group [L] int) with the same residue
indexing for model and native; `group` is the chain (1 or 2; 0 = residue not in the file).  Atom slot 1 is CA
(constants.BBHeavyAtom.CA).

Pinning: Fnat, the contact counts and the interface list are checked against the reference's own `fnat` binary built from its
sources (oracle/Makefile -> oracle/_ref/fnat) on PDB files written from the same tensors (tests/golden/make_golden.py::case_dockq).
The superposition part (Bio.PDB.Superimposer / SVDSuperimposer, DockQ.py:296-353) is the textbook SVD (Kabsch) fit with the
reflection correction Biopython applies (flip the last right-singular vector when det < 0).  Biopython is not installed here, so that
part is pinned to published algorithms instead: `kabsch` (SVD) and `qcp` below (Theobald 2005 / Liu, Agrafiotis & Theobald 2010: the
largest root of the quartic characteristic polynomial of the 4x4 key matrix by Newton iteration, rotation from the adjugate of
K - lambda I) are two INDEPENDENT derivations of the same optimum -- no shared linear-algebra routine -- and must agree on every
fixture, including near-degenerate fits (collinear CA atoms, 3-residue interfaces) and mirror-image candidates
(tests/test_oracle_golden.py::test_dockq_superposition_two_algorithms); scipy's Rotation.align_vectors is a third check in the golden
generator.  What stays unpinned is only Biopython's own floating-point path, not the optimum it computes.
"""
import numpy as np

CA = 1


def residue_min_dist2(pos, mask):
    """(L, L) minimum squared heavy-atom distance between residues (molecule.c:581-612); inf where a residue has no atom."""
    pos = np.asarray(pos, np.float64)
    L, A, _ = pos.shape
    d2 = ((pos[:, None, :, None, :] - pos[None, :, None, :, :]) ** 2).sum(-1)            # (L, L, A, A)
    ok = mask[:, None, :, None] & mask[None, :, None, :]
    return np.where(ok, d2, np.inf).reshape(L, L, A * A).min(-1)


def contacts(pos, mask, group, cutoff):
    """Residue pairs (a in chain 1, b in chain 2) with min distance <= cutoff (fnat.c:100-160: `crd(m,i,j) <= cutoff` on squared values)."""
    md2 = residue_min_dist2(pos, mask)
    g = np.asarray(group)
    return (md2 <= cutoff * cutoff) & (g[:, None] == 1) & (g[None, :] == 2)


def kabsch(x, y):
    """Rotation R and translation t minimising |R y + t - x| (Bio.SVDSuperimposer: y = model atoms moved onto x = reference)."""
    cx, cy = x.mean(0), y.mean(0)
    H = (y - cy).T @ (x - cx)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    D = np.diag([1.0, 1.0, d])
    R = Vt.T @ D @ U.T
    return R, cx - R @ cy


def qcp(x, y, iters=100, tol=1e-15):
    """Quaternion characteristic polynomial superposition (Theobald, Acta Cryst. A61 (2005) 478; Liu, Agrafiotis & Theobald,
    J. Comput. Chem. 31 (2010) 1561): -> (rmsd, R, t) with the same convention as `kabsch` (R y + t ~ x), WITHOUT an SVD or an
    eigen-solver.  lambda_max of the key matrix is the largest root of  l^4 + C2 l^2 + C1 l + C0  (C2 = -2 tr(S^T S), C1 = -8 det S,
    C0 = det K), found by Newton's iteration from the upper bound E0 = (|x|^2 + |y|^2) / 2; the optimal quaternion is any non-zero
    column of adj(K - lambda I)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    n = x.shape[0]
    cx, cy = x.mean(0), y.mean(0)
    xc, yc = x - cx, y - cy
    S = yc.T @ xc                                    # S[a, b] = sum y_a x_b
    Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz = S.reshape(-1)
    K = np.array([[Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx],
                  [Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz],
                  [Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy],
                  [Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz]])
    E0 = 0.5 * ((xc ** 2).sum() + (yc ** 2).sum())
    C2 = -2.0 * (S ** 2).sum()
    C1 = -8.0 * (Sxx * (Syy * Szz - Syz * Szy) - Sxy * (Syx * Szz - Syz * Szx) + Sxz * (Syx * Szy - Syy * Szx))

    def det3(m):
        return (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
                + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))

    def minor(m, i, j):
        return det3(np.delete(np.delete(m, i, 0), j, 1))
    C0 = sum((-1) ** j * K[0, j] * minor(K, 0, j) for j in range(4))
    lam = E0
    for _ in range(iters):
        l2 = lam * lam
        p = l2 * l2 + C2 * l2 + C1 * lam + C0
        dp = 4 * l2 * lam + 2 * C2 * lam + C1
        if dp == 0:
            break
        step = p / dp
        lam -= step
        if abs(step) < tol * max(abs(lam), 1.0):
            break
    rmsd = float(np.sqrt(max(0.0, 2.0 * (E0 - lam) / n)))
    M = K - lam * np.eye(4)
    adj = np.array([[(-1) ** (i + j) * minor(M, j, i) for j in range(4)] for i in range(4)])
    q = adj[:, np.argmax((adj ** 2).sum(0))]          # the column of largest norm: a multiple of the eigenvector
    nq = np.linalg.norm(q)
    if nq < 1e-300:                                   # lambda_max is a multiple root (degenerate fit): any proper rotation attaining it
        return rmsd, None, None
    a, b, c, d = q / nq
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])
    return rmsd, R, cx - R @ cy


def rmsd_after_fit(x, y):
    R, t = kabsch(x, y)
    return float(np.sqrt((((y @ R.T + t) - x) ** 2).sum(-1).mean()))


def dockq(model_pos, model_mask, native_pos, native_mask, group):
    """-> dict(fnat, nat_correct, nat_total, irms, Lrms, DockQ) for ONE model (DockQ.py:98-385 with use_CA_only=True)."""
    group = np.asarray(group)
    nat5 = contacts(native_pos, native_mask, group, 5.0)
    mod5 = contacts(model_pos, model_mask, group, 5.0)
    nat_total = int(nat5.sum())
    nat_correct = int((nat5 & mod5).sum())
    fnat = nat_correct / nat_total if nat_total else 0.0                                  # fnat.c:238-243
    nat10 = contacts(native_pos, native_mask, group, 10.0)                                # the interface comes from the NATIVE contacts at 10 A (DockQ.py:110,121-123)
    interface = nat10.any(1) | nat10.any(0)
    both_ca = model_mask[:, CA] & native_mask[:, CA] & (group > 0)                        # atoms_def_in_both, CA only (DockQ.py:150-188)
    sel = interface & both_ca
    x, y = np.asarray(native_pos, np.float64)[:, CA], np.asarray(model_pos, np.float64)[:, CA]
    irms = rmsd_after_fit(x[sel], y[sel])                                                 # DockQ.py:296-301
    n1, n2 = int((both_ca & (group == 1)).sum()), int((both_ca & (group == 2)).sum())
    # receptor = the chain with MORE common atoms; chain1 (first in the file) is the ligand unless it is strictly longer (DockQ.py:303-318)
    rec, lig = (1, 2) if n1 > n2 else (2, 1)
    rsel, lsel = both_ca & (group == rec), both_ca & (group == lig)
    R, t = kabsch(x[rsel], y[rsel])                                                       # align on the receptor (DockQ.py:330-332)
    Lrms = float(np.sqrt((((y[lsel] @ R.T + t) - x[lsel]) ** 2).sum(-1).mean()))           # ligand RMSD without refitting (DockQ.py:358-366)
    score = (fnat + 1 / (1 + (irms / 1.5) ** 2) + 1 / (1 + (Lrms / 8.5) ** 2)) / 3         # DockQ.py:378
    return dict(fnat=fnat, nat_correct=nat_correct, nat_total=nat_total, irms=irms, Lrms=Lrms, DockQ=score,
                interface=interface, n_interface=int(sel.sum()))


# ------------------------------------------------------------------ PDB text (golden generation / pinning against the reference binary)
_AA3 = ['ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU', 'MET', 'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR', 'UNK']
_ATOM_NAMES = ['N', 'CA', 'C', 'O', 'CB'] + [f'X{k}' for k in range(10)]


def write_pdb(path, pos, mask, group, aa=None, chain_ids='AB'):
    """Minimal PDB writer: residue number = index + 1, chain = chain_ids[group - 1]; residues in chain order (1 then 2)."""
    lines, serial = [], 1
    for g in (1, 2):
        for i in np.nonzero(np.asarray(group) == g)[0]:
            res = _AA3[int(aa[i])] if aa is not None else 'ALA'
            for a in range(pos.shape[1]):
                if not mask[i, a]:
                    continue
                x, y, z = (float(v) for v in pos[i, a])
                name = _ATOM_NAMES[a]
                lines.append('ATOM  %5d %-4s %3s %s%4d    %8.3f%8.3f%8.3f  1.00  0.00           %s' % (
                    serial, (' ' + name) if len(name) < 4 else name, res, chain_ids[g - 1], i + 1, x, y, z, name[0] if name[0] in 'NCO' else 'C'))
                serial += 1
        lines.append('TER')
    lines.append('END')
    with open(path, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


def parse_reference_fnat(text):
    """Fields of the reference binary's stdout the reference itself parses (DockQ.py:18-49)."""
    import re
    out = dict(inter=set())
    for line in text.split('\n'):
        m = re.search(r'NATIVE: (\d+)(\w) (\d+)(\w)', line)
        if line.startswith('Fnat'):
            f = line.split(' ')
            out.update(nat_correct=int(f[1]), nat_total=int(f[2]), fnat=float(f[3]))
        elif line.startswith('Fnonnat'):
            f = line.split(' ')
            out.update(nonnat_count=int(f[1]), model_total=int(f[2]))
        elif m:
            out['inter'].add((int(m.group(1)), m.group(2)))
            out['inter'].add((int(m.group(3)), m.group(4)))
    return out
