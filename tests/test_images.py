"""CPU: image ingest (src/utils/load-images.ts) -- file filtering and ordering, PNG/JPEG decode to rgba8, PNG writer."""
import io
import os

import numpy as np
import pytest

from webdgs_amd import images
from webdgs_amd.viewer import encodePNG


def _frame(h=37, w=53, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    a[::3, :, :3] = (np.arange(w, dtype=np.uint8) * 3)[None, :, None]  # smooth rows so encoders pick non-trivial filters
    return a


def test_png_writer_round_trips_through_both_decoders():
    a = _frame()
    data = encodePNG(a)
    assert np.array_equal(images.decodePNG(data), a)
    assert np.array_equal(images.decodeImage(data), a)
    with pytest.raises(ValueError):
        encodePNG(a[..., :3])
    with pytest.raises(ValueError):
        images.decodePNG(b"not a png at all")


def test_fallback_png_decoder_handles_every_filter_and_colour_type():
    Image = pytest.importorskip("PIL.Image")
    a = _frame(29, 41, 1)
    cases = {
        "RGBA": a,
        "RGB": a[..., :3],
        "L": a[..., 0],
        "LA": a[..., [0, 3]],
    }
    for mode, arr in cases.items():
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(arr), mode).save(buf, format="PNG", optimize=True)
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.uint8)
        assert np.array_equal(images.decodePNG(buf.getvalue()), want), mode
    pal = Image.fromarray(a[..., :3], "RGB").quantize(16)
    buf = io.BytesIO()
    pal.save(buf, format="PNG")
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.uint8)
    assert np.array_equal(images.decodePNG(buf.getvalue()), want)


def test_natural_order_matches_numeric_locale_compare():
    names = ["img10.png", "img2.png", "IMG1.PNG", "img02b.png", "a.png", "B.png"]
    assert sorted(names, key=images.naturalKey) == ["a.png", "B.png", "IMG1.PNG", "img2.png", "img02b.png", "img10.png"]


def test_load_images_filters_sorts_and_drops_undecodable_files(tmp_path, capsys):
    a, b = _frame(8, 12, 2), _frame(8, 12, 3)
    files = {"frame_10.png": encodePNG(a), "frame_9.PNG": encodePNG(b), "notes.txt": b"hello", "broken.png": b"\x89PNG\r\n\x1a\nxxxx"}
    try:
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(a[..., :3], "RGB").save(buf, format="JPEG", quality=95)
        files["frame_11.jpg"] = buf.getvalue()
    except ImportError:
        pass
    for name, data in files.items():
        with open(os.path.join(tmp_path, name), "wb") as f:
            f.write(data)
    got = images.loadImages([os.path.join(tmp_path, n) for n in files], device=None)
    names = [im.name for im in got]
    assert names[:2] == ["frame_9.PNG", "frame_10.png"]  # numeric, case-insensitive; broken.png dropped, notes.txt filtered
    assert "broken.png" not in names and "notes.txt" not in names
    assert "Failed to load image broken.png" in capsys.readouterr().err
    assert np.array_equal(got[0].bitmap, b) and np.array_equal(got[1].bitmap, a)
    assert (got[0].width, got[0].height) == (12, 8) and got[0].texture is None
    if "frame_11.jpg" in files:
        assert names[2] == "frame_11.jpg" and got[2].bitmap.shape == (8, 12, 4) and (got[2].bitmap[..., 3] == 255).all()
