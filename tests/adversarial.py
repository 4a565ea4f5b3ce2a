"""Scenes aimed at the distance culling of the default traversal (a subtree is skipped when its box is entered
beyond closest_hit * (1 + 2^-9) + 2^-13; the reference, accel/bvh.c:106-111, skips nothing): slivers whose
Moeller-Trumbore determinant sits just above the 1e-7 rejection threshold (ray.c:44-50), meshes far from the
origin, a camera inside a mesh, rays grazing the zero-thickness boxes of a flat mesh, and a large triangle
hit at a grazing angle behind a near blocker.  Used on the CPU (host build of the pixel code) and on the GPU.
Every scene is a list of (triangle soup [k,24] f32) per mesh, placements, camera, frame size."""
import numpy as np


def _tri(v0, v1, v2):
    r = np.zeros(24, np.float32)
    r[0:3], r[3:6], r[6:9] = v0, v1, v2
    r[9:15] = (0.1, 0.1, 0.9, 0.1, 0.5, 0.9)
    n = np.cross(np.subtract(v1, v0), np.subtract(v2, v0))
    ln = np.linalg.norm(n)
    n = n / ln if ln > 0 else np.array([0.0, 1.0, 0.0])
    r[15:18] = r[18:21] = r[21:24] = n
    return r


def _blob(rng, center, radius, n, size=0.3):
    rows = []
    for _ in range(n):
        c = np.asarray(center) + rng.normal(size=3) * radius
        rows.append(_tri(c, c + rng.normal(size=3) * radius * size, c + rng.normal(size=3) * radius * size))
    return rows


def _basis(rng):
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    a = np.cross(d, rng.normal(size=3))
    a /= np.linalg.norm(a)
    return d, a, np.cross(d, a)


def scenes(po, seed=0):
    """Yields (name, soups, transforms, camera, width, height)."""
    rng = np.random.default_rng(seed)
    ident = po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))
    default_cam = po.Camera(po.vec3(-3, 3, -3), po.vec3(0, 0, 0), po.vec3(0, 1, 0), 40.0)

    # slivers: long thin triangles; |a| = |e1 . (d x e2)| ~ length * width * cos, swept across the 1e-7 threshold
    for width in (3e-7, 1e-6, 1e-5, 1e-3):
        rows = []
        for _ in range(300):
            c = rng.normal(size=3) * 0.6
            d, e, _ = _basis(rng)
            length = 10 ** rng.uniform(-1, 0.5)
            rows.append(_tri(c, c + d * length, c + d * length * 0.5 + e * width / length))
        yield f"slivers width {width:g}", [rows + _blob(rng, np.zeros(3), 0.6, 100)], [ident], default_cam, 160, 90

    # far from the origin: coordinates of 1e3 .. 1e4 in the triangles themselves and in the placement
    for off in (1.0e3, 3.3e3, 1.0e4):
        o = np.array([off, 0.5 * off, -off])
        cam = po.Camera(po.vec3(*(o + [-3, 3, -3])), po.vec3(*o), po.vec3(0, 1, 0), 60.0)
        plane = [_tri([-50, -1, -50], [50, -1, 50], [50, -1, -50]), _tri([-50, -1, -50], [-50, -1, 50], [50, -1, 50])]
        yield (f"far {off:g}", [_blob(rng, o, 0.7, 300), plane],
               [ident, po.Transform(po.vec3(o[0], o[1], o[2]), po.vec3(0.1, 0.7, 0.0))], cam, 160, 90)

    # camera inside a closed-ish cloud of triangles, looking outwards through it
    cam = po.Camera(po.vec3(0.05, 0.02, -0.03), po.vec3(1, 0.3, 0.5), po.vec3(0, 1, 0), 90.0)
    shell = []
    for _ in range(600):
        d, a, b = _basis(rng)
        c = d * rng.uniform(0.5, 2.5)
        shell.append(_tri(c + a * 0.25, c - a * 0.2 + b * 0.2, c - b * 0.25))
    yield "camera inside", [shell], [ident], cam, 160, 90

    # grazing rays along a flat mesh (boxes of zero thickness), other geometry just above it
    n = 12
    xs = np.linspace(-6, 6, n + 1)
    flat = []
    for j in range(n):
        for i in range(n):
            p = lambda a_, b_: [xs[a_], 0.0, xs[b_]]                     # noqa: E731
            flat += [_tri(p(i, j), p(i + 1, j + 1), p(i + 1, j)), _tri(p(i, j), p(i, j + 1), p(i + 1, j + 1))]
    for hgt in (1e-1, 1e-3, 1e-5):
        cam = po.Camera(po.vec3(-5.5, hgt, 0.013), po.vec3(5.0, 0.0, 0.4), po.vec3(0, 1, 0), 30.0)
        yield f"grazing flat mesh height {hgt:g}", [flat, _blob(rng, [0, 0.05, 0], 1.5, 200, size=0.1)], [ident, ident], cam, 160, 90
        yield (f"grazing flat mesh height {hgt:g}, one mesh", [flat + _blob(rng, [0, 0.05, 0], 1.5, 200, size=0.1)], [ident], cam, 160, 90)

    # a triangle in general position hit at a grazing angle g (its distance then carries a relative error of
    # about 3e-7 / g) behind a blocker that faces the camera; the view is as narrow as the grazing band
    for k in range(6):
        dist = 10 ** rng.uniform(0.7, 2.0)
        size = 10 ** rng.uniform(-0.5, 0.4)
        g = 10 ** rng.uniform(-6.5, -4.0)
        d, a, b = _basis(rng)
        pos = rng.normal(size=3)
        center = pos + d * dist
        dp = d + g * b
        dp /= np.linalg.norm(dp)
        rows = [_tri(center - dp * size - a * size, center + dp * size - a * size * 0.2, center + a * size)]
        delta = (1.2 * size + 0.004 * dist) * rng.uniform(1.0, 1.5)
        bc = pos + d * (dist - delta)
        rows.append(_tri(bc - a * size - b * size, bc + a * size - b * size, bc + b * size))
        for _ in range(30):
            c = center + a * rng.uniform(6, 18) * size * rng.choice([-1, 1]) + b * rng.uniform(-3, 3) + d * rng.uniform(-2, 2)
            rows.append(_tri(c, c + rng.normal(size=3) * 0.2, c + rng.normal(size=3) * 0.2))
        fov = float(np.degrees(2 * np.arctan(4 * size * g / dist)))
        cam = po.Camera(po.vec3(*pos), po.vec3(*center), po.vec3(*b), fov)
        yield f"grazing triangle g={g:.2g} behind a blocker", [rows], [ident], cam, 96, 54

    # attributes the reference never checks: normals of zero length and NaN normals (the shaded normal is 0/0, the shadow ray
    # starts at a NaN point and meets nothing, fmaxf(0.2f, NaN) keeps the ambient term), texture coordinates far outside
    # [0, 1), negative, and so large that u - floorf(u) is 0
    rows = _blob(rng, np.zeros(3), 0.8, 240, size=0.5)
    for k, r in enumerate(rows):
        if k % 4 == 0:
            r[15:24] = 0.0
        elif k % 4 == 1:
            r[15:18] = np.nan
        elif k % 4 == 2:
            r[9:15] = (-3.25, 17.5, 1.0e9, -1.0e9, 123456.78, -0.001)
        else:
            r[9:15] = (1.0e30, -1.0e30, 1.0, 1.0, 0.0, 0.0)
    yield "zero and NaN normals, texture coordinates far outside the unit square", [rows], [ident], default_cam, 160, 90


def _beam(a, b, width, up):
    """A box of twelve triangles around the segment a-b: `width` wide across the segment in the plane normal to `up`, and
    as tall along `up`."""
    a, b, up = np.asarray(a, float), np.asarray(b, float), np.asarray(up, float)
    along = (b - a) / np.linalg.norm(b - a)
    side = np.cross(up, along)
    side *= 0.5 * width / np.linalg.norm(side)
    rise = up * (0.5 * width / np.linalg.norm(up))
    c = [p + sx * side + sy * rise for p in (a, b) for sx in (-1, 1) for sy in (-1, 1)]      # index = 4 end + 2 side + rise
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    rows = []
    for q in quads:
        rows.append(_tri(c[q[0]], c[q[1]], c[q[2]]))
        rows.append(_tri(c[q[0]], c[q[2]], c[q[3]]))
    return rows


def diagonal_frames(po, seed=0):
    """Meshes of long thin beams along the diagonals of a coordinate plane -- what the packer gives a tree in a sheared frame
    (Packer::build_shear) -- seen from where that tree's conditions hold and from where they do not: a general view, views
    whose rays run inside the planes of the beams' faces (those faces' triangles are tested apart), a camera inside the
    hull of the widened triangles and one touching a beam (the other trees), one beyond the tree's reach, views along the
    axes.  Yields (name, soups, transforms, camera, light, width, height, kept axis of the expected frame)."""
    rng = np.random.default_rng(seed)
    light = po.default_light()
    axes = np.eye(3)
    for kept in (1, 0, 2):
        up = axes[kept]
        a, b = axes[(kept + 1) % 3], axes[(kept + 2) % 3]
        rows = []
        for off in np.linspace(-0.6, 0.6, 7):
            rows += _beam(-(a + b) + off * (a - b) * 0.5, (a + b) + off * (a - b) * 0.5, 0.03, up)
            rows += _beam(-(a - b) + off * (a + b) * 0.5, (a - b) + off * (a + b) * 0.5, 0.02, up)
        rows += _blob(rng, up * 0.2, 0.15, 40, size=0.2)                    # a hub of small triangles facing every way
        ident = po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))
        turned = po.Transform(po.vec3(0.3, 0.2, -0.1), po.vec3(0.2, 0.7, -0.4))
        floor = [_tri(-3 * (a + b) - 1.5 * up, 3 * (a - b) - 1.5 * up, 3 * (a + b) - 1.5 * up),
                 _tri(-3 * (a + b) - 1.5 * up, 3 * (a + b) - 1.5 * up, -3 * (a - b) - 1.5 * up)]
        soups = [rows, floor]

        def cam(pos, at=(0, 0, 0), fov=50.0):
            u = up if abs(np.dot(np.subtract(at, pos), up)) < 0.99 * np.linalg.norm(np.subtract(at, pos)) else a
            return po.Camera(po.vec3(*pos), po.vec3(*at), po.vec3(*u), fov)
        views = [("general view", ident, cam(2.0 * up + 1.3 * a + 0.4 * b)),
                 ("placed and turned", turned, cam(2.5 * up - 1.0 * a + 2.0 * b)),
                 ("view inside the planes of one diagonal's faces", ident, cam(1.5 * up + 1.5 * (a + b), fov=30.0)),
                 ("view inside the planes of the other diagonal's faces", ident, cam(1.0 * up + 2.0 * (a - b), fov=30.0)),
                 ("view along the beams' tops", ident, cam(0.015 * up + 3.0 * a, at=0.015 * up, fov=40.0)),
                 ("camera between the beams", ident, cam(0.05 * up + 0.1 * a, at=2.0 * a + 1.0 * b)),
                 ("camera on a beam's face", ident, cam(0.0151 * up + 0.3 * (a + b), at=0.5 * up + a)),
                 ("camera just above the beams", ident, cam(0.3 * up, at=a + 0.2 * b, fov=90.0)),
                 ("camera beyond the tree's reach", ident, cam(9.0 * up + 6.0 * a + 2.0 * b, fov=15.0)),
                 ("view along the kept axis", ident, cam(3.0 * up, fov=40.0)),
                 ("view along a sheared axis", ident, cam(3.0 * a, fov=40.0))]
        for name, tf, c in views:
            yield f"beams in the plane normal to axis {kept}: {name}", soups, [tf, ident], c, light, 128, 80, kept
        for ldir in ((1, 1, 1), tuple(a + b + 0.001 * up), tuple(up), tuple(0.7 * up + a - b)):
            l = po.lib().orc_make_light(po.vec3(*ldir), po.vec3(1.0, 0.9, 0.8))
            yield f"beams in the plane normal to axis {kept}: light {ldir}", soups, [turned, ident], cam(2.0 * up + 1.3 * a + 0.4 * b), l, 128, 80, kept


def build(po, soups, texture):
    """Oracle-side meshes (reference build of every soup)."""
    out = []
    for s in soups:
        t, nodes = po.build_bvh(np.asarray(s, np.float32))
        out.append(po.HostMesh(t, nodes, texture))
    return out


def degenerate_views(po):
    """Yields (name, transforms, camera, light) for the reference driver's three meshes: inputs on which the reference's arithmetic
    runs into zeros, infinities and NaNs -- a camera looking at its own position or along its up vector (the basis of
    camera.c:13-15 is 0/0), fields of view of 0, 180 and beyond, a light direction of zero length or straight along an axis,
    placements far away, with NaN angles or infinite offsets.  The reference reports no errors for any of them (its result is
    whatever IEEE arithmetic gives, mostly the background); a replacement has to give the same bytes."""
    nan, inf = float("nan"), float("inf")
    tf = po.animation_transforms(17)
    cam = po.default_camera()
    light = po.default_light()

    def camera(pos=(-3, 3, -3), at=(0, 0, 0), up=(0, 1, 0), fov=60.0):
        return po.Camera(po.vec3(*pos), po.vec3(*at), po.vec3(*up), fov)

    def raw_light(d, c=(1.4, 1.4, 1.4)):
        return po.Light(po.vec3(*d), po.vec3(*c))

    def moved(k, pos=None, rot=None):
        out = list(tf)
        out[k] = po.Transform(po.vec3(*(pos if pos is not None else tf[k].position.tup())), po.vec3(*(rot if rot is not None else tf[k].rotation.tup())))
        return out

    yield "camera at its own look-at point", tf, camera(at=(-3, 3, -3)), light
    yield "camera looking along its up vector", tf, camera(pos=(0, 4, 0), at=(0, 0, 0), up=(0, 1, 0)), light
    yield "up vector of zero length", tf, camera(up=(0, 0, 0)), light
    yield "up vector of 1e30 (the normalisation overflows: right = up = 0, every pixel gets the forward ray)", tf, camera(up=(0, 1, 1e30)), light
    yield "up vector of 1e-20 (the normalisation runs into denormals)", tf, camera(up=(0, 1e-20, 0)), light
    yield "up vector of 1e-30", tf, camera(up=(0, 1e-30, 0)), light
    yield "look-at point 1e-20 away from the camera", tf, camera(pos=(0, 1, -2), at=(1e-20, 1, -2 + 1e-20)), light
    yield "camera 1e19 away", tf, camera(pos=(-3e19, 3e19, -3e19)), light
    yield "field of view 0", tf, camera(fov=0.0), light
    yield "field of view 180", tf, camera(fov=180.0), light
    yield "field of view 270", tf, camera(fov=270.0), light
    yield "negative field of view", tf, camera(fov=-60.0), light
    yield "NaN camera position", tf, camera(pos=(nan, 3, -3)), light
    yield "camera infinitely far away", tf, camera(pos=(-inf, 3, -3)), light
    yield "light direction of zero length", tf, cam, raw_light((0, 0, 0))
    yield "light straight down an axis (two zero components)", tf, cam, raw_light((0, 1, 0))
    yield "light from below", tf, cam, raw_light((0, -1, 0))
    yield "light direction not normalised", tf, cam, raw_light((3, 4, -2))
    yield "NaN light direction", tf, cam, raw_light((nan, 1, 0))
    yield "light direction of 1e30 (its normalisation in mesh space overflows: shadow rays without a direction)", tf, cam, raw_light((-1e30, 0.2, -0.85))
    yield "light direction of 1e-25 (its square leaves the float range downwards)", tf, cam, raw_light((1e-25, 2e-25, -1e-25))
    yield "negative light colour", tf, cam, raw_light(light.direction.tup(), (-1.0, 0.5, 300.0))
    yield "drone with a NaN angle", moved(0, rot=(0, nan, 0)), cam, light
    yield "treasure infinitely far away", moved(1, pos=(inf, 0, 0)), cam, light
    yield "treasure at 1e30", moved(1, pos=(1e30, 0, 0)), cam, light
    yield "ground with an infinite angle", moved(2, rot=(inf, 0, 0)), cam, light
    yield "camera inside the treasure's box, axis-parallel view", moved(1, pos=(0, 0, 0), rot=(0, 0, 0)), camera(pos=(0, 0.5, 0), at=(0, 0.5, 5)), light
    yield "camera exactly on the ground plane, looking along it", tf, camera(pos=(-3, 0, 0), at=(3, 0, 0)), light


def nonfinite_soups(seed=1, sizes=(7, 60, 500, 5000)):
    """Yields (name, soup [n,24] f32): triangles with NaN and infinite vertices, coordinates whose sums overflow, denormals, all
    triangles identical, flat sets.  The reference checks nothing (mesh.c:40-76 reads whatever sscanf gives): its box folding
    ignores NaNs and whatever lies beyond +-1e30 (aabb.c:3-23), a NaN centroid makes every comparison of the partition false
    (bvh.c:36-49), and the walk's slab and triangle tests reject NaNs; builders and kernels have to end up with the same bits."""
    rng = np.random.default_rng(seed)
    for n in sizes:
        base = rng.normal(size=(n, 24)).astype(np.float32) * np.float32(0.7)
        s = base.copy(); s[rng.integers(n), 0] = np.nan; yield f"one NaN coordinate, n={n}", s
        s = base.copy(); s[::3, :9] = np.nan; yield f"every third triangle NaN, n={n}", s
        s = base.copy(); s[n // 2:, :9] = np.nan; yield f"second half NaN, n={n}", s
        s = base.copy(); s[:, :9] = np.nan; yield f"all NaN, n={n}", s
        s = base.copy(); s[rng.integers(n), 4] = np.inf; yield f"one infinite coordinate, n={n}", s
        s = base.copy(); s[::2, 1] = -np.inf; s[1::4, 1] = np.inf; yield f"infinities of both signs, n={n}", s
        s = base.copy(); s[:, 0] = s[:, 3] = s[:, 6] = np.inf; yield f"all x infinite, n={n}", s
        with np.errstate(over="ignore"):
            s = base.copy(); s[:, :9] *= np.float32(3e38); yield f"sums that overflow, n={n}", s
        s = base.copy(); s[:, :9] *= np.float32(1e-42); yield f"denormals, n={n}", s
        s = base.copy(); s[:, :9] = s[0, :9]; yield f"all triangles identical, n={n}", s
        s = base.copy(); s[:, 0] = s[:, 3] = s[:, 6] = np.float32(1.0); yield f"flat in x, n={n}", s
