"""Polarimetric film: a polarimetric sensor stores the 4 Stokes components per channel
(include/wt/sensor/film/film.hpp:214-286 with FilmSampleT = vec4).  CPU checks through the CPU checker; the GPU parity test is marked."""
import numpy as np
import pytest

from oracle_util import oracle_render

CASES = [("cornell_box", dict(res=24, mesh_detail=0, crop_of=1440, lut=(64, 64)), 8),
         ("etoile", dict(res=48, mesh_detail=0), 16),
         ("furnace_path", dict(res=16), 4),
         # BASELINE.json configs[4] stand-in: scenes/bidir_room/room.xml, polarimetric wave mode
         ("bidir_room", dict(res=60, mesh_detail=0, lut=(128, 128)), 8)]


def _planes(sc, v, w, l, spp):
    from wave_tracer_amd import develop
    img = develop(sc, v, w, l, spp).astype(np.float64)
    return img.reshape(img.shape[0], img.shape[1], sc.spectral_channels, sc.stokes)


@pytest.mark.parametrize("name,kw,spp", CASES)
def test_stokes_film_intensity_plane_and_degree_of_polarization(built, name, kw, spp):
    from wave_tracer_amd import Scene
    a, b = Scene(name, **kw), Scene(name, polarimetric=1, **kw)
    assert (a.stokes, b.stokes) == (1, 4) and b.channels == 4 * a.channels
    ia = _planes(a, *oracle_render(a, 0, spp, 3)[:3], spp)
    ib = _planes(b, *oracle_render(b, 0, spp, 3)[:3], spp)
    # the I plane is the intensity image, bit for bit (same samples, same arithmetic)
    assert np.array_equal(ib[..., 0], ia[..., 0])
    assert ia.sum() > 0
    # every pixel is a sum of physical Stokes vectors expressed in the sensor's frame: degree of polarisation <= 1
    pol = np.sqrt((ib[..., 1:] ** 2).sum(axis=-1))
    assert (pol <= ib[..., 0] * (1 + 1e-4) + 1e-30).all()
    if name == "cornell_box":
        # dielectric and conductor reflections polarise: linear components are a sizeable fraction, circular is tiny
        assert np.abs(ib[..., 1]).sum() > 0.02 * ib[..., 0].sum()
        assert np.abs(ib[..., 3]).sum() < 0.01 * ib[..., 0].sum()
    if name == "furnace_path":
        # Lambertian surfaces depolarise completely
        assert np.abs(ib[..., 1:]).sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,spp", CASES)
def test_stokes_film_gpu_parity(built, name, kw, spp):
    """All four Stokes planes of the HIP path against the CPU checker: relative L1 per plane group (fp32 arithmetic)."""
    from wave_tracer_amd import Scene, render
    sc = Scene(name, polarimetric=1, **kw)
    g = _planes(sc, *render(sc, spp, seed=3, device=0), spp)
    c = _planes(sc, *oracle_render(sc, 0, spp, 3)[:3], spp)
    assert np.isfinite(g).all()
    assert np.abs(g[..., 0] - c[..., 0]).sum() <= 2e-2 * np.abs(c[..., 0]).sum()
    assert np.abs(g[..., 1:] - c[..., 1:]).sum() <= 3e-2 * max(np.abs(c[..., 1:]).sum(), 1e-3 * np.abs(c[..., 0]).sum())
