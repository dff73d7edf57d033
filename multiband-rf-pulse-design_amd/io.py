"""Pulse file writers: rfwrite_varian.m, rfwrite.m and rf_tools/signa.m, host side.
Same file contents as the reference's fprintf / fwrite calls; the interactive `input('Root file name: ')` becomes the
`root_fname` argument (an empty name means 'Not saving files', as there)."""
import math

import numpy as np

GAMMA_H1 = 4257.0          # Hz/G (rfwrite.m:25)


def _mround(v):
    """MATLAB round: halves away from zero."""
    v = np.asarray(v, dtype=np.float64)
    return np.sign(v) * np.floor(np.abs(v) + 0.5)


def signa(wav, fn, s=None):
    """signa.m:21-60: 16-bit integers with the low bit masked off (an odd value would read as end-of-sequence),
    full scale 32766 unless a scale is given; a complex waveform goes to fn + '.r' / '.i'.  Written with the machine's
    byte order, like MATLAB's fopen(fn, 'w') / fwrite(..., 'short')."""
    wav = np.asarray(wav).ravel()
    wmax = 0x7FFE
    if s is None:
        s = 1.0 / max(np.max(np.abs(wav.real)), np.max(np.abs(wav.imag)) if np.iscomplexobj(wav) else 0.0)
    wav = wav * s * wmax
    if np.iscomplexobj(wav):
        wav = 2 * _mround(wav.real / 2) + 2j * _mround(wav.imag / 2)
        if np.sum(np.abs(wav.imag)) == 0:
            wav = wav.real
    else:
        wav = 2 * _mround(wav / 2)
    def put(name, v):
        with open(name, "wb") as f:
            f.write(np.clip(v, -32768, 32767).astype("=i2").tobytes())      # fwrite saturates
    if np.iscomplexobj(wav):
        put(fn + ".r", wav.real)
        put(fn + ".i", wav.imag)
    else:
        put(fn, wav)


def rfwrite_varian(rf, nompw, nombw=None, root_fname=None):
    """rfwrite_varian.m:21-59: Varian/Agilent .RF text file (phase in degrees, magnitude scaled to 1024, gate 1).
    nompw in ms, nombw in kHz (default 0.5, unused by the scanner)."""
    if nombw is None:
        nombw = 0.5
    if not root_fname:
        return None                                                        # 'Not saving files'
    rf = np.asarray(rf, dtype=np.complex128).ravel()
    pha = np.angle(rf) * 180 / np.pi
    mag = 1024 * np.abs(rf) / np.max(np.abs(rf))
    integral = np.sum(mag) / (1024 * len(mag))
    name = "%s.RF" % root_fname
    with open(name, "w") as f:
        f.write("# VERSION   100\n# TYPE    selective\n# MODULATION  amplitude\n# EXCITEWIDTH   1.8125\n# INVERTWIDTH   0\n")
        f.write("# INTEGRAL   %1.5f\n" % integral)
        f.write("# T(ms)xBW(KHz)   %1.4fx%2.1f\n" % (nompw, nombw))
        for p, m in zip(pha, mag):
            f.write("%3.7f \t %4.7f \t %2.7f\n" % (p, m, 1))
    return name


def rfwrite(rf, nompw, ang, GAMMA=None, isodelay=None, g=None, thk=None, root_fname=None):
    """rfwrite.m:23-141: GE `.dat` statistics file plus `.rho` (and `.pha` for a complex pulse, `.grd` when a
    gradient and a thickness are given).  rf in G, nompw in s, ang in radians, GAMMA in Hz/G (default protons);
    the sixth argument is the nominal bandwidth in Hz when no thickness follows (rfwrite.m:78-86)."""
    if GAMMA is None:
        GAMMA = GAMMA_H1
    gwrite = thk is not None
    if not root_fname:
        return None
    rf = np.asarray(rf).ravel() * GAMMA / GAMMA_H1
    maxrf = np.max(np.abs(rf))
    rfn = rf / maxrf
    nrf = len(rf)
    pon = np.real(rfn) >= 0.00001                                          # MATLAB compares the real part
    temp_pw = max_pw = 0
    for v in pon:
        temp_pw += int(v)
        if not v and temp_pw != 0:
            max_pw = max(max_pw, temp_pw)
            temp_pw = 0
    max_pw = max_pw / nrf
    dty_cyc = max(np.sum(np.abs(rfn) > 0.2236) / nrf, max_pw)
    if g is None:
        nombw, maxg = 0.0, None
    elif gwrite:
        maxg = float(np.max(np.abs(g)))
        nombw = GAMMA * maxg * thk
    else:
        nombw, maxg = float(g), None
    name = "%s.dat" % root_fname
    with open(name, "w") as f:
        f.write("%10d \t\t #extgradfile\n" % int(gwrite))
        f.write("%10d \t\t #res\n" % nrf)
        f.write("%10d \t\t #pw\n" % int(_mround(nompw * 1e6)))
        f.write("%10.7f \t\t #nom_flip \n" % (ang * 180 / math.pi))
        f.write("%10.7f \t\t #abswidth \n" % (np.sum(np.abs(rfn)) / nrf))
        f.write("%10.7f \t\t #effwidth \n" % (np.sum(np.abs(rfn) ** 2) / nrf))
        f.write("%10.7f \t\t #area \n" % (np.sum(np.abs(rfn)) / nrf))
        f.write("%10.7f \t\t #dtycyc \n" % dty_cyc)
        f.write("%10.7f \t\t #maxpw \n" % max_pw)
        f.write("%10.7f \t\t #max_b1 \n" % maxrf)
        f.write("%10.7f \t\t #max_int_b1_sqr \n" % np.sum(np.abs(rf) ** 2 * nompw / nrf * 1e3))
        f.write("%10.7f \t\t #max_rms_b1 \n" % (math.sqrt(np.sum(np.abs(rf) ** 2)) / nrf))
        f.write("%10.7f \t\t #nom_bw \n" % nombw)
        if gwrite:
            f.write("%10.3f \t\t #a_gzs \n" % maxg)
            f.write("%10.3f \t\t #nom_thk(mm) \n" % (thk * GAMMA / GAMMA_H1 * 10))
    if np.iscomplexobj(rf) and np.any(rf.imag):
        signa(np.abs(rfn), "%s.rho" % root_fname)
        signa(np.angle(rfn), "%s.pha" % root_fname, 1 / math.pi)
    else:
        signa(np.real(rfn), "%s.rho" % root_fname)
    if gwrite:
        signa(g, "%s.grd" % root_fname)
    return name
