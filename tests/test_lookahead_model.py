"""tools/lookahead_model.py (host-only model of the cull with a lookahead window, no GPU and no library): the shares it predicted before the engine
was built -- and that profiles/r06_lookahead.md then measured -- keep their order: knowing more of the keyframes that follow never renders more."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rendered_share_falls_with_the_window():
    spec = importlib.util.spec_from_file_location("lookahead_model", os.path.join(ROOT, "tools", "lookahead_model.py"))
    lm = importlib.util.module_from_spec(spec); spec.loader.exec_module(lm)
    shares = {(D, mode): lm.run(60, D, mode, first=20) for D in (0, 4, 24) for mode in ("all", "tile", "free")}
    for mode in ("all", "tile", "free"):
        r0, r4, r24 = shares[(0, mode)][0], shares[(4, mode)][0], shares[(24, mode)][0]
        assert 0.35 < r0 < 0.7 and r4 < 0.85 * r0 and r24 <= r4, (mode, r0, r4, r24)
    # a fresh tile that may be left fresh (the form that was built) renders no more than one its first keyframe must render, and no less than
    # cells of fresh tiles culled one by one; without lookahead the three are the same map
    assert shares[(0, "all")] == shares[(0, "tile")] == shares[(0, "free")]
    assert shares[(4, "free")][0] <= shares[(4, "tile")][0] <= shares[(4, "all")][0]
    for v in shares.values():
        assert v[0] <= v[1] <= v[2] <= 1.0                      # rendered cells, within one cell of them, within two
