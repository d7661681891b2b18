"""Shared helpers of the parity tests: render one scene on the oracle and on the HIP path and compare every target."""
import numpy as np

TARGET_NAMES = ["depth", "scene_color", "gbuffer_a", "gbuffer_b", "gbuffer_c", "gbuffer_d"]


def render_both(pyoracle, engine, cfg, shadow_dim=1024, flags=0, debug_view=0, loader=None):
    from zeldaengine_amd import engine as eng
    W, H = cfg["width"], cfg["height"]
    o = pyoracle.Oracle(W, H, shadow_dim)
    (loader or pyoracle.load_scene)(o, cfg)
    o.render(debug_view)
    g = engine.Renderer(W, H, shadow_dim, flags=flags, debug_view=debug_view)
    (loader or eng.load_scene)(g, cfg)
    g.render()
    g.finish()
    return o, g


def compare_all(o, g, check_color=True):
    """Returns a dict name -> number of differing elements (bit-exact comparison on the raw words)."""
    diffs = {}
    so, sg = o.shadowmap(), g.shadowmap()
    diffs["shadowmap"] = int((so.view(np.uint32) != sg.view(np.uint32)).sum())
    for t, name in enumerate(TARGET_NAMES):
        a, b = o.gbuffer(t), g.gbuffer(t)
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        diffs[name] = int((a != b).sum())
    if check_color:
        diffs["color"] = int((o.color() != g.color()).any(axis=2).sum())
    return diffs
