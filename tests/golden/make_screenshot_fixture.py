#!/usr/bin/env python3
"""tests/golden/make_screenshot_fixture.py -- turn a screenshot the reference's README shows into a small test fixture.

/root/reference/img/panini.webp is a window capture of the reference program itself (monoportal scene, Panini projection,
the camera panel open: look-at (0,0,0), alpha -116.7 deg, beta 70.3 deg, R 1.80, parameter 1.0, view angle 220 deg).  It is
the only output of the real renderer this build can be held against (no Rust, no GL here), so it is kept -- cropped to the
client area the scene is drawn into and reduced to 262 x 188 -- together with the parameters read off the panel.  The capture
is older than the scene files (the walls were more saturated then, the ceiling tiles lighter), lossy and partly covered by the
GUI, so tests/test_reference_screenshot.py compares WHERE things are (hue classes), not pixel values.

Run in the build container (needs /root/reference and PIL):  python tests/golden/make_screenshot_fixture.py
"""
import json
import os

from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_screenshots")
SIZE = (262, 188)

if __name__ == "__main__":
    shot = Image.open("/root/reference/img/panini.webp").convert("RGB")
    w, h = shot.size
    client = (10, 44, w - 9, h - 10)  # inside the window frame, below the title bar: what draw_texture(0, 0, screen_w, screen_h) covers
    crop = shot.crop(client)
    cw, ch = crop.size
    crop.resize(SIZE, Image.BOX).save(os.path.join(OUT, "panini.png"))
    sx, sy = SIZE[0] / cw, SIZE[1] / ch
    covered = [  # GUI drawn over the scene, in fixture pixels: x0, y0, x1, y1
        [0, 0, SIZE[0], int((80 - 44) * sy) + 2],                                                       # menu bar
        [int((30 - 10) * sx), int((98 - 44) * sy), int((548 - 10) * sx) + 1, int((482 - 44) * sy) + 1],  # camera panel
    ]
    meta = {
        "source": "img/panini.webp of the reference repository (README screenshot of the running program)",
        "client_size": [cw, ch],
        "scene": "monoportal",
        "uniforms": {"triangle_x": -0.5},
        "camera": {"look_at": [0.0, 0.0, 0.0], "alpha_deg": -116.7, "beta_deg": 70.3, "r": 1.80},
        "options": {"use_panini_projection": 1, "panini_param": 1.0, "view_angle_deg": 220.0},
        "covered": covered,
    }
    json.dump(meta, open(os.path.join(OUT, "panini.json"), "w"), indent=1)
    print("wrote", OUT, crop.size, "->", SIZE)

    # img/interface.webp: the program's editor on an (almost) empty scene -- the only thing drawn is the gizmo of the identity matrix
    # `id` (DebugMatrix capsules: x red, y green, z blue) on the sky colour, with the camera panel open (look-at (0, 0.61, 0),
    # alpha 44.9, beta 56.4, R 4.24, view angle 90).  Kept: the part of the client area between the panels, at half size.
    shot = Image.open("/root/reference/img/interface.webp").convert("RGB")
    w, h = shot.size
    crop = shot.crop((10, 44, w - 9, h - 10))
    cw, ch = crop.size
    half = crop.resize((cw // 2, ch // 2), Image.BOX)
    box = [(700 - 10) // 2, (475 - 44) // 2, (1150 - 10) // 2, (765 - 44) // 2]  # x0, y0, x1, y1 in half-size client pixels
    half.crop(box).save(os.path.join(OUT, "interface_gizmo.png"))
    meta = {
        "source": "img/interface.webp of the reference repository (README screenshot of the running program)",
        "render_size": [cw // 2, ch // 2],
        "box": box,
        "camera": {"look_at": [0.0, 0.61, 0.0], "alpha_deg": 44.9, "beta_deg": 56.4, "r": 4.24},
        "view_angle_deg": 90.0,
    }
    json.dump(meta, open(os.path.join(OUT, "interface_gizmo.json"), "w"), indent=1)
    print("wrote interface_gizmo", box)

    # img/monoportal.webp: the same scene in the ordinary projection, control panel open (triangle_x -0.5, y 0, z -1) but NO camera
    # panel.  The three numbers of the orbit camera were fitted (coarse grid + coordinate descent on the hue-class agreement,
    # look-at left at the origin): alpha 258.0, beta 62.5 degrees, R 2.67.  Three parameters cannot make a wrong renderer agree on
    # ~50 000 pixels, but it is a fit and says so.
    shot = Image.open("/root/reference/img/monoportal.webp").convert("RGB")
    w, h = shot.size
    crop = shot.crop((10, 44, w - 9, h - 10))
    cw, ch = crop.size
    size = (cw // 6, ch // 6)
    crop.resize(size, Image.BOX).save(os.path.join(OUT, "monoportal.png"))
    sx, sy = size[0] / cw, size[1] / ch
    meta = {
        "source": "img/monoportal.webp of the reference repository (README screenshot of the running program)",
        "client_size": [cw, ch],
        "scene": "monoportal",
        "uniforms": {"triangle_x": -0.5},
        "camera": {"look_at": [0.0, 0.0, 0.0], "alpha_deg": 258.0, "beta_deg": 62.5, "r": 2.67},
        "camera_fitted": True,
        "options": {"view_angle_deg": 90.0},
        "covered": [[0, 0, size[0], int((80 - 44) * sy) + 2],
                    [int((50 - 10) * sx), int((130 - 44) * sy), int((575 - 10) * sx) + 1, int((580 - 44) * sy) + 1]],
    }
    json.dump(meta, open(os.path.join(OUT, "monoportal.json"), "w"), indent=1)
    print("wrote monoportal", size)

    # img/mobius_monoportal.webp: the Moebius-band portal (a Complex object whose intersection is a numerical search written in the
    # scene file), stage "Explore", no camera panel: alpha 88.5, beta 71.0 degrees, R 3.475 fitted the same way.
    shot = Image.open("/root/reference/img/mobius_monoportal.webp").convert("RGB")
    w, h = shot.size
    crop = shot.crop((10, 44, w - 9, h - 10))
    cw, ch = crop.size
    size = (cw // 6, ch // 6)
    crop.resize(size, Image.BOX).save(os.path.join(OUT, "mobius_monoportal.png"))
    sx, sy = size[0] / cw, size[1] / ch
    meta = {
        "source": "img/mobius_monoportal.webp of the reference repository (README screenshot of the running program)",
        "client_size": [cw, ch],
        "scene": "mobius_monoportal",
        "stage": "explore",
        "uniforms": {},
        "camera": {"look_at": [0.0, 0.0, 0.0], "alpha_deg": 88.5, "beta_deg": 71.0, "r": 3.475},
        "camera_fitted": True,
        "options": {"view_angle_deg": 90.0},
        "covered": [[0, 0, size[0], int((80 - 44) * sy) + 2],
                    [int((30 - 10) * sx), int((98 - 44) * sy), int((548 - 10) * sx) + 1, int((760 - 44) * sy) + 1]],
    }
    json.dump(meta, open(os.path.join(OUT, "mobius_monoportal.json"), "w"), indent=1)
    print("wrote mobius_monoportal", size)
