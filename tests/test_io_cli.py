"""File IO (CPU) and command-line front ends (GPU) -- SURVEY 8(f) rows 3-4,
modelled on the reference's tests/run_denoising_test.py:18-79 and
tests/run_deconvolution_test.py:18-73 (smoke: exit status 0) plus value checks
against the solver API."""
import os

import numpy as np
import pytest

from conftest import rel_l2


def test_nifti_round_trip_and_reader_writer(tmp_path, golden):
    from nsol_amd import nifti
    from nsol_amd.data_reader import DataReader
    from nsol_amd.data_writer import DataWriter
    vol = golden("configs")["phantom64"].astype(np.float64)
    p = str(tmp_path / "vol.nii.gz")
    nifti.write(p, vol.astype(np.float32), (1.0, 2.0, 0.5))
    arr, spacing, hdr = nifti.read(p)
    assert np.array_equal(arr, vol) and spacing == (1.0, 2.0, 0.5)
    r = DataReader(p)
    r.read_data()
    assert np.array_equal(r.get_data(), vol)
    assert r.get_image_sitk().GetSpacing() == (1.0, 2.0, 0.5)
    # result written "next to" its input keeps the header / spacing
    out = str(tmp_path / "sub" / "res.nii.gz")
    DataWriter(vol * 2, out, r.get_image_sitk()).write_data()
    arr2, spacing2, _ = nifti.read(out)
    assert np.allclose(arr2, vol * 2) and spacing2 == (1.0, 2.0, 0.5)
    for ext in ("npy", "mat", "png"):
        q = str(tmp_path / ("img." + ext))
        img = golden("configs")["lena_noise_u8"].astype(np.float64)
        DataWriter(img, q).write_data()
        rr = DataReader(q)
        rr.read_data()
        assert np.array_equal(rr.get_data(), img)
        assert rr.get_image_sitk() is None
    DataWriter(np.arange(6.).reshape(2, 3), str(tmp_path / "a.txt")).write_data()
    assert os.path.getsize(str(tmp_path / "a.txt")) > 0
    with pytest.raises(IOError):
        DataReader(str(tmp_path / "missing.png")).read_data()
    nifti.write(str(tmp_path / "u8.nii"), np.arange(24, dtype=np.uint8)
                .reshape(2, 3, 4))
    a, s, _ = nifti.read(str(tmp_path / "u8.nii"))
    assert a.shape == (2, 3, 4) and a[1, 2, 3] == 23 and s == (1., 1., 1.)


def test_cli_requires_result():
    from nsol_amd.application import run_denoising
    with pytest.raises(IOError):
        run_denoising.main(["--observation", "x.png"])


@pytest.mark.gpu
def test_similarity_measures_vs_numpy():
    """similarity_measures.py:26-120 formulas, evaluated by nsol_pair_stats."""
    from nsol_amd.similarity_measures import SimilarityMeasures as sm
    rng = np.random.default_rng(4)
    x = 100.0 + 30.0 * rng.standard_normal((17, 23, 9))
    r = x + 5.0 * rng.standard_normal(x.shape)
    n = float(x.size)
    assert np.isclose(sm.sum_of_squared_differences(x, r),
                      np.sum(np.square(x - r)), rtol=1e-12)
    assert np.isclose(sm.sum_of_absolute_differences(x, r),
                      np.sum(np.abs(x - r)), rtol=1e-12)
    assert np.isclose(sm.mean_absolute_error(x, r),
                      np.sum(np.abs(x - r)) / n, rtol=1e-12)
    mse = np.sum(np.square(x - r)) / n
    assert np.isclose(sm.mean_squared_error(x, r), mse, rtol=1e-12)
    assert np.isclose(sm.root_mean_square_error(x, r), np.sqrt(mse),
                      rtol=1e-12)
    assert np.isclose(sm.peak_signal_to_noise_ratio(x, r),
                      10 * np.log10(np.max(r) ** 2 / mse), rtol=1e-12)
    ncc = np.sum((x - x.mean()) * (r - r.mean())) / \
        float(x.size * x.std(ddof=1) * r.std(ddof=1))
    assert np.isclose(sm.normalized_cross_correlation(x, r), ncc, rtol=1e-11)
    # identities of tests/similarity_measures_test.py:32-94
    assert sm.sum_of_squared_differences(x, x) == 0
    assert np.isclose(sm.normalized_cross_correlation(x, x),
                      (n - 1) / n, rtol=1e-12)
    with pytest.raises(ValueError):
        sm.mean_squared_error(x, r[:-1])
    assert set(sm.similarity_measures) >= {"SSD", "MAE", "MSE", "RMSE",
                                           "PSNR", "NCC"}


@pytest.mark.gpu
@pytest.mark.parametrize("rtype", ["TVL1", "TVL2", "HuberL1", "HuberL2"])
def test_run_denoising_cli_2d_and_3d(tmp_path, golden, rtype):
    from nsol_amd import nifti
    from nsol_amd.data_writer import DataWriter
    from nsol_amd.data_reader import DataReader
    from nsol_amd.application import run_denoising
    g = golden("configs")
    png = str(tmp_path / "2D_Lena_256_noise.png")
    DataWriter(g["lena_noise_u8"].astype(np.float64), png).write_data()
    nii = str(tmp_path / "3D_SheppLoganPhantom_64.nii.gz")
    nifti.write(nii, g["phantom64"].astype(np.float64))
    for src, ext in ((png, "png"), (nii, "nii.gz")):
        out = str(tmp_path / ("out_%s.%s" % (rtype, ext)))
        rc = run_denoising.main(["--observation", src, "--result", out,
                                 "--reconstruction-type", rtype,
                                 "--iterations", "5", "--reference", src,
                                 "--dtype", "float64"])
        assert rc == 0 and os.path.isfile(out)
        r = DataReader(out)
        r.read_data()
        obs = DataReader(src)
        obs.read_data()
        s = run_denoising.build_solver(obs.get_data(), rtype, 0.03, 5,
                                       dtype=np.float64)
        s.run()
        want = s.get_x().reshape(obs.get_data().shape)
        if ext == "png":
            want = np.round(want).astype(np.uint8)
            assert np.array_equal(r.get_data(), want)
        else:
            assert rel_l2(r.get_data(), want) < 1e-6      # float32 file


@pytest.mark.gpu
def test_run_denoising_cli_reproduces_config1(tmp_path, golden):
    from nsol_amd.data_writer import DataWriter
    from nsol_amd.data_reader import DataReader
    from nsol_amd.application import run_denoising
    g = golden("configs")
    src = str(tmp_path / "lena.npy")
    DataWriter(g["lena_noise_u8"].astype(np.float64), src).write_data()
    out = str(tmp_path / "recon.npy")
    assert run_denoising.main(["--observation", src, "--result", out,
                               "--iterations", "50", "--alpha", "0.03"]) == 0
    r = DataReader(out)
    r.read_data()
    assert rel_l2(r.get_data(), g["cfg1_lena_TVL2_50it_L2eq8"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rtype,solver", [("TK0L2", "PD"), ("TK1L2", "PD"),
                                          ("TVL2", "PD"), ("TVL2", "ADMM"),
                                          ("HuberL2", "PD")])
def test_run_deconvolution_cli(tmp_path, golden, rtype, solver):
    from nsol_amd import nifti
    from nsol_amd.data_writer import DataWriter
    from nsol_amd.application import run_deconvolution
    g = golden("configs")
    png = str(tmp_path / "lena.png")
    DataWriter(g["lena_noise_u8"][:96, :128].astype(np.float64), png) \
        .write_data()
    nii = str(tmp_path / "vol.nii.gz")
    nifti.write(nii, g["phantom64"][:24, :32, :40].astype(np.float64),
                (1.0, 1.0, 2.0))
    for src, ext in ((png, "png"), (nii, "nii.gz")):
        out = str(tmp_path / ("dec_%s_%s.%s" % (rtype, solver, ext)))
        rc = run_deconvolution.main([
            "--observation", src, "--result", out, "--blur", "1.2",
            "--reconstruction-type", rtype, "--solver", solver,
            "--iterations", "3", "--iter-max", "4"])
        assert rc == 0 and os.path.getsize(out) > 0
