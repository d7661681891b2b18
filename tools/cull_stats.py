"""Where the camera pass's culls drop their meshlet-instances (config 3 / 4 / 5): python tools/cull_stats.py [3|4|5 ...]"""
import sys
sys.path.insert(0, '/root/repo')
from zeldaengine_amd import engine, scenes
for c in [int(x) for x in sys.argv[1:]] or [3, 4]:
    cfg = scenes.config3(10000) if c == 3 else scenes.config4(1000000, 256 if c == 5 else 16)
    g = engine.Renderer(cfg["width"], cfg["height"], 1024)
    engine.load_scene(g, cfg)
    for i in range(6):
        g.update_uniforms(cfg["camera"], cfg["dir"], cfg["point"], cfg["spot"], 0.0, 0.002 * i, 0.016 * i)
        g.render()
    g.finish()
    st = g.stats()
    sel = st["hiz_culled"] - st["hiz_culled_geom"]
    print("config %d: work items %d, camera survivors (reach k_geom, both rounds, minus k_geom's own drops) %d, round 1 %d; Hi-Z culled %d = k_select (box bounds, before any vertex work) %d + k_geom (after the vertex phase) %d; shadow survivors %d occluded %d late %d"
          % (c, st["work_items"][1], st["survivors"][1], st["round1_survivors"], st["hiz_culled"], sel, st["hiz_culled_geom"], st["survivors"][0], st["shadow_occluded"], st["shadow_late"]))
    g.close()
