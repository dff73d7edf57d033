"""Pulse file writers (mbfir.io) against the reference's fprintf / fwrite statements, evaluated by hand."""
import numpy as np

import mbfir


def test_rfwrite_varian_text(tmp_path):
    rf = np.array([0.1, 0.2j, -0.4, 0.2 - 0.2j])
    name = mbfir.rfwrite_varian(rf, 2.32, None, str(tmp_path / "p"))
    lines = open(name).read().split("\n")
    assert lines[:5] == ["# VERSION   100", "# TYPE    selective", "# MODULATION  amplitude", "# EXCITEWIDTH   1.8125",
                         "# INVERTWIDTH   0"]
    mag = 1024 * np.abs(rf) / 0.4
    assert lines[5] == "# INTEGRAL   %1.5f" % (mag.sum() / (1024 * 4))          # rfwrite_varian.m:47,54
    assert lines[6] == "# T(ms)xBW(KHz)   2.3200x0.5"                            # default nombw 0.5 kHz (:23)
    assert lines[7] == "0.0000000 \t 256.0000000 \t 1.0000000"
    assert lines[8] == "90.0000000 \t 512.0000000 \t 1.0000000"
    assert lines[9] == "180.0000000 \t 1024.0000000 \t 1.0000000"
    assert lines[10] == "-45.0000000 \t %4.7f \t 1.0000000" % (1024 * np.sqrt(0.08) / 0.4)
    assert lines[11] == "" and len(lines) == 12
    assert mbfir.rfwrite_varian(rf, 2.32, None, "") is None                       # 'Not saving files'


def test_signa_masks_the_low_bit_and_scales(tmp_path):
    fn = str(tmp_path / "w.rho")
    mbfir.signa(np.array([1.0, 0.5, -0.25, 0.00004, 1.0 / 32766 * 3]), fn)
    v = np.fromfile(fn, dtype="=i2")
    assert list(v) == [32766, 16384, -8192, 2, 4]                                 # 16383 -> 2 round(8191.5) = 16384; 1.31 -> 2; 3 -> 4
    mbfir.signa(np.array([np.pi / 2, -np.pi]), fn, 1 / np.pi)                     # phase file: scale 1/pi (rfwrite.m:126)
    assert list(np.fromfile(fn, dtype="=i2")) == [16384, -32766]
    mbfir.signa(np.array([1 + 0.5j, -0.5j]), fn)                                  # complex: two files
    assert list(np.fromfile(fn + ".r", dtype="=i2")) == [32766, 0]
    assert list(np.fromfile(fn + ".i", dtype="=i2")) == [16384, -16384]


def test_rfwrite_dat_and_waveforms(tmp_path):
    rf = np.array([0.0, 0.05, 0.1, 0.05, 0.0, -0.02, 0.0])
    root = str(tmp_path / "sat")
    name = mbfir.rfwrite(rf, 0.0035, np.pi / 2, None, 0, None, None, root)
    lines = [ln.split("\t")[0].strip() for ln in open(name).read().strip().split("\n")]
    tags = [ln.split("#")[1].strip() for ln in open(name).read().strip().split("\n")]
    assert tags == ["extgradfile", "res", "pw", "nom_flip", "abswidth", "effwidth", "area", "dtycyc", "maxpw", "max_b1",
                    "max_int_b1_sqr", "max_rms_b1", "nom_bw"]
    rfn = rf / 0.1
    assert lines[0] == "0" and lines[1] == "7" and lines[2] == "3500" and lines[3] == "90.0000000"
    assert lines[4] == "%.7f" % (np.abs(rfn).sum() / 7) and lines[5] == "%.7f" % ((rfn ** 2).sum() / 7) and lines[6] == lines[4]
    assert lines[8] == "%.7f" % (3 / 7)                                           # longest run of rfn >= 1e-5: samples 2-4
    assert lines[7] == "%.7f" % (3 / 7)                                           # |rfn| > 0.2236: 3 samples = maxpw
    assert lines[9] == "0.1000000" and lines[12] == "0.0000000"
    assert lines[10] == "%.7f" % np.sum(rf ** 2 * 0.0035 / 7 * 1e3) and lines[11] == "%.7f" % (np.sqrt(np.sum(rf ** 2)) / 7)
    assert list(np.fromfile(root + ".rho", dtype="=i2")) == [0, 16384, 32766, 16384, 0, -6554, 0]
    # complex pulse on another nucleus with a gradient: magnitude + phase + gradient files, two more .dat lines
    rfc = rf * np.exp(0.5j)
    name = mbfir.rfwrite(rfc, 0.0035, np.pi / 2, 1070.5, 0, np.array([0.0, 1.0, 2.0, 1.0]), 0.5, root)
    txt = open(name).read()
    assert txt.startswith("         1 \t\t #extgradfile\n") and "#a_gzs" in txt and "#nom_thk(mm)" in txt
    assert "%10.7f \t\t #nom_bw \n" % (1070.5 * 2.0 * 0.5) in txt
    assert "%10.7f \t\t #max_b1 \n" % (0.1 * 1070.5 / 4257) in txt                # scaled by GAMMA / GAMMA_H1 (rfwrite.m:42)
    assert list(np.fromfile(root + ".grd", dtype="=i2")) == [0, 16384, 32766, 16384]
    pha = np.fromfile(root + ".pha", dtype="=i2")
    assert pha[2] == 2 * round(0.5 / np.pi * 32766 / 2) and len(pha) == 7
