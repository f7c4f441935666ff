"""The .ray reader on damaged input: a scene or an error, never a crash.

The reference's parser throws ParserException / SyntaxErrorException on bad input and RayTracer::loadScene turns
that into "could not load" (src/parser/Parser.cpp, src/RayTracer.cpp:254-268).  The host front end must do the
same for whatever text it is given: every mutant of the test scenes below either loads (and flattens into a
scene description) or raises.  tools/sanitize_host.sh runs this file against an AddressSanitizer build too."""
import random

from raytracer_amd import host, scenes
from tests import common

TOKENS = ["{", "}", "(", ")", ",", ";", "=", "1e999", "-", "nan", "inf", "0", "-1", "1.5", "polymesh", "points", "faces",
          "materials", "normals", "gennormals", "sphere", "box", "cone", "cylinder", "square", "translate", "rotate",
          "scale", "transform", "material", "camera", "point_light", "spot_light", "directional_light",
          "ambient_light", "SBT-raytracer", "1.0", "map(\"x.png\")", "\"", "/*", "*/", "//", "\n", "\0", "capped",
          "height", "top_radius", "bottom_radius", "index", "shininess"]


def mutate(rnd, text):
    s = list(text)
    for _ in range(rnd.choice((1, 1, 2, 3, 8))):
        op, pos = rnd.random(), rnd.randrange(0, len(s) + 1)
        if op < 0.3 and s:
            del s[pos:pos + rnd.choice((1, 1, 2, 5, 40))]
        elif op < 0.6:
            s[pos:pos] = list(rnd.choice(TOKENS))
        elif op < 0.75 and s:
            s[min(pos, len(s) - 1)] = chr(rnd.randrange(1, 127))
        elif op < 0.85:
            s = s[:pos]
        else:
            a = rnd.randrange(0, len(s) + 1)
            s[pos:pos] = s[a:min(len(s), a + rnd.choice((5, 50, 500)))]
    return "".join(s)


def test_mutated_scene_files_load_or_raise(built):
    seeds = [scenes.simple_scene(), common.all_primitives_scene(), common.small_mesh_scene(1),
             common.vertex_material_scene(), scenes.cornell_scene()]
    rnd = random.Random(2024)
    loaded = rejected = 0
    for _ in range(4000):
        text = mutate(rnd, rnd.choice(seeds))
        try:
            hs = host.HostScene(text=text)
            assert hs.desc.n_objects >= 0 and hs.desc.n_lights >= 0
            loaded += 1
        except Exception:
            rejected += 1
    # both outcomes occur: the mutations are neither all harmless nor all fatal
    assert loaded > 100 and rejected > 1000
