"""Host-side scene inputs of the NRC-HPM path (values only; no Vulkan plumbing).

Mirrors what the reference uploads as UBOs/textures before a frame:
  * camera          src/Camera.cpp:164-174 (glm::perspective * glm::lookAt, inverse), defaults src/main.cu:180-187
  * scene presets   src/AppConfig.cpp:93-150 (HpmSceneConfig ids 0-5)
  * lights          src/HpmScene.cpp:28-30, src/DirLight.cpp:5-14
  * volume box      src/NrcHpmRenderer.cu:910-912 (normalize(extent) * 107.5), g = 0.8 src/HpmScene.cpp:45
  * density texel   src/Texture3D.cpp:106 (uint8(v*255) truncation), stored here as ONE channel (R8)
Synthetic volumes / env map follow SURVEY.md section 8(d) (seeded, deterministic).
"""
import math

import numpy as np

SCENE_PRESETS = {  # id: (dirLightStrength, pointLightStrength, hdrEnvMapStrength, density)
    0: (16.0, 0.0, 0.0, 0.6),
    1: (0.0, 64.0, 0.0, 0.6),
    2: (0.0, 128.0, 0.0, 1.0),
    3: (16.0, 0.0, 0.0, 0.25),
    4: (8.0, 0.0, 0.1, 0.6),
    5: (0.0, 0.0, 1.0, 1.6),
}


def perspective(fovy, aspect, near, far):
    """glm::perspective, GL clip space (depth -1..1), right handed."""
    t = math.tan(fovy / 2.0)
    m = np.zeros((4, 4), np.float64)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = -(far + near) / (far - near)
    m[3, 2] = -1.0
    m[2, 3] = -(2.0 * far * near) / (far - near)
    return m


def look_at(eye, center, up):
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m


def make_camera(pos=(64.0, 0.0, 0.0), view_dir=(-1.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), aspect=1920.0 / 1080.0,
                fovy=math.radians(60.0), near=0.1, far=100.0):
    """CameraMatrices.invProjView (column-major, as glm stores it) + camera.pos."""
    pos = np.asarray(pos, np.float64)
    pv = perspective(fovy, aspect, near, far) @ look_at(pos, pos + np.asarray(view_dir, np.float64), up)
    inv = np.linalg.inv(pv)
    return dict(inv_proj_view=np.ascontiguousarray(inv.T.astype(np.float32).reshape(16)),  # column-major
                pos=pos.astype(np.float32))


def dir_light_dir(zenith=-1.57, azimuth=0.0):
    """VecFromAngles (src/DirLight.cpp:5-14): Ry(azimuth) * Rx(zenith) * (0,1,0)."""
    cz, sz = math.cos(zenith), math.sin(zenith)
    v = np.array([0.0, cz, sz])
    ca, sa = math.cos(azimuth), math.sin(azimuth)
    return np.array([ca * v[0] + sa * v[2], v[1], -sa * v[0] + ca * v[2]], np.float32)


def volume_size(dims):
    """skySize = normalize(vec3(extent)) * 107.5 in fp32 (src/NrcHpmRenderer.cu:910-912)."""
    d = np.asarray(dims, np.float32)
    n = np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2])))
    return (d / n * np.float32(107.5)).astype(np.float32)


def quantize_density(vol_f32):
    """src/Texture3D.cpp:106: uint8(value * 255.0f) (truncation). Input indexed [i][j][k] = (x,y,z)."""
    q = (np.asarray(vol_f32, np.float32) * np.float32(255.0)).astype(np.uint8)
    # memory order of the texture: index = i + W*j + W*H*k  -> array [k][j][i]
    return np.ascontiguousarray(q.transpose(2, 1, 0))


def white_env(value=1.0):
    """Quirk Q9 (src/read_file.cpp:129-130): every loaded env texel is overwritten with 1.0."""
    return np.full((1, 1, 4), value, np.float32)


def black_env():
    """Empty hdrEnvMapPath -> 1x1 black (src/read_file.cpp:85-90)."""
    return np.zeros((1, 1, 4), np.float32)


def procedural_sky(w=512, h=256):
    """Synthetic HDR lat-long env map: vertical gradient + sun lobe."""
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    theta = (v - 0.5) * np.float32(math.pi)          # asin(y) range
    phi = (u - 0.5) * np.float32(2 * math.pi)        # atan(z, x) range
    y = np.sin(theta)[:, None] * np.ones((1, w), np.float32)
    x = np.cos(theta)[:, None] * np.cos(phi)[None, :]
    z = np.cos(theta)[:, None] * np.sin(phi)[None, :]
    up = np.clip(y * 0.5 + 0.5, 0, 1)
    sky = (0.25 + 0.75 * up)[..., None] * np.array([0.55, 0.75, 1.0], np.float32)
    sun_dir = np.array([0.0, 0.5, 0.8660254], np.float32)
    c = np.clip(x * sun_dir[0] + y * sun_dir[1] + z * sun_dir[2], 0, 1)
    sun = (c ** 256)[..., None] * np.array([40.0, 36.0, 30.0], np.float32) + (c ** 8)[..., None] * 0.6
    env = np.concatenate([sky + sun, np.ones((h, w, 1), np.float32)], axis=2)
    return np.ascontiguousarray(env.astype(np.float32))


def sphere_volume(n=64):
    """C1 plumbing volume: d(p) = clamp(1 - |p-c|/r, 0, 1), r = 0.45 n."""
    g = (np.arange(n, dtype=np.float32) + 0.5) - n / 2.0
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    d = 1.0 - np.sqrt(x * x + y * y + z * z) / np.float32(0.45 * n)
    d = np.clip(d, 0, 1).astype(np.float32)
    d /= d.max()
    return d


def _value_noise3(n, cells, rng):
    """Trilinear value noise on an n^3 grid from a (cells+1)^3 random lattice."""
    lat = rng.random((cells + 1,) * 3, dtype=np.float32)
    t = np.linspace(0, cells, n, endpoint=False, dtype=np.float32)
    i0 = np.floor(t).astype(np.int32)
    f = t - i0
    f = f * f * (3 - 2 * f)
    i1 = np.minimum(i0 + 1, cells)

    def ax(a, i, axis):
        return np.take(a, i, axis=axis)

    fx, fy, fz = f[:, None, None], f[None, :, None], f[None, None, :]
    c = 0
    for dx, wx in ((i0, 1 - fx), (i1, fx)):
        a = ax(lat, dx, 0)
        for dy, wy in ((i0, 1 - fy), (i1, fy)):
            b = ax(a, dy, 1)
            for dz, wz in ((i0, 1 - fz), (i1, fz)):
                c = c + ax(b, dz, 2) * (wx * wy * wz)
    return c.astype(np.float32)


def fbm_cloud_volume(n=256, seed=1337, octaves=5):
    """C2/C3 cloud: fBm-perturbed ellipsoid, max normalised to exactly 1.0 (src/Texture3D.cpp:74)."""
    rng = np.random.default_rng(seed)
    noise = np.zeros((n, n, n), np.float32)
    amp, cells, tot = 1.0, 4, 0.0
    for _ in range(octaves):
        noise += amp * _value_noise3(n, cells, rng)
        tot += amp
        amp *= 0.5
        cells *= 2
    noise /= tot
    g = (np.arange(n, dtype=np.float32) + 0.5) / n * 2 - 1
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    r = np.sqrt((x / 0.85) ** 2 + (y / 0.6) ** 2 + (z / 0.8) ** 2)
    d = np.clip((1.0 - r) * 1.6 + (noise - 0.5) * 1.8, 0, None)
    d = np.clip(d * 2.5, 0, 1).astype(np.float32)
    d /= d.max()
    return d


def smoke_volume(n=512, seed=1337):
    """C5 heterogeneous smoke plume (seeded)."""
    rng = np.random.default_rng(seed)
    noise = np.zeros((n, n, n), np.float32)
    amp, cells, tot = 1.0, 4, 0.0
    for _ in range(4):
        noise += amp * _value_noise3(n, cells, rng)
        tot += amp
        amp *= 0.5
        cells *= 2
    noise /= tot
    g = (np.arange(n, dtype=np.float32) + 0.5) / n * 2 - 1
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    hgt = (y + 1) * 0.5
    rad = np.sqrt((x - 0.3 * np.sin(4 * hgt)) ** 2 + z ** 2)
    d = np.clip((0.15 + 0.5 * hgt - rad) * 3.0, 0, None) * np.clip(noise * 2.2 - 0.6, 0, None)
    d = np.clip(d, 0, 1).astype(np.float32)
    d /= d.max()
    return d


def cached_volume(kind, n, seed=1337, cache_dir="/tmp"):
    """quantised synthetic volume (`kind`: "cloud" = fbm_cloud_volume, "smoke" = smoke_volume, "sphere"), generated once per
    box and kept as an .npy under `cache_dir` (the 512^3 smoke takes about a minute and 5 GB of host memory to generate)"""
    import os
    gen = {"cloud": fbm_cloud_volume, "smoke": smoke_volume, "sphere": lambda n, seed=0: sphere_volume(n)}[kind]
    path = os.path.join(cache_dir, "nrc_%s_%d_%d.npy" % (kind, n, seed))
    if os.path.exists(path):
        return np.load(path)
    vol = quantize_density(gen(n, seed=seed))
    try:
        tmp = "%s.%d.tmp.npy" % (path, os.getpid())
        np.save(tmp, vol)
        os.replace(tmp, path)
    except OSError:
        pass
    return vol


def make_scene(density_u8, scene_id=4, env=None, g=0.8, dims=None, size=None):
    """Bundle scene inputs.  density_u8: uint8 array in texture memory order [k][j][i] (see quantize_density)."""
    dl, pl, env_s, rho = SCENE_PRESETS[scene_id]
    density_u8 = np.ascontiguousarray(density_u8, np.uint8)
    if dims is None:
        nz, ny, nx = density_u8.shape
        dims = (nx, ny, nz)
    if env is None:
        env = white_env() if env_s != 0.0 else black_env()
    return dict(
        density=density_u8, dims=tuple(int(d) for d in dims),
        size=np.asarray(size if size is not None else volume_size(dims), np.float32),
        density_factor=float(np.float32(rho)), g=float(np.float32(g)),
        dir_light_dir=dir_light_dir(), dir_light_strength=float(dl),
        point_light_pos=np.zeros(3, np.float32), point_light_strength=float(pl),
        point_light_color=np.ones(3, np.float32),
        env=np.ascontiguousarray(env, np.float32), env_strength=float(np.float32(env_s)),
        scene_id=scene_id,
    )


class HpmScene:
    """Host mirror of en::HpmScene (src/HpmScene.cpp:22-100): scene preset + the directional light's angles, with the
    reference's per-frame `Update(deltaTime)`.  Only preset 3 moves when `dynamic` is set (its directional light's azimuth
    advances by deltaTime/2, wrapped at 2*3.141 -- the literal of src/HpmScene.cpp:68); the reference ships every preset with
    dynamic = false (src/AppConfig.cpp:96-150), so this is off unless asked for.  `scene` is the dict make_scene returns;
    hand it to `NrcHpmRenderer.SetSceneParams` / `McHpmRenderer.SetSceneParams` after an update."""

    def __init__(self, density_u8, scene_id=4, env=None, dynamic=False, **kw):
        self.scene = make_scene(density_u8, scene_id=scene_id, env=env, **kw)
        self.scene_id = scene_id
        self.dynamic = bool(dynamic)
        self.zenith, self.azimuth = -1.57, 0.0       # src/HpmScene.cpp:28

    def IsDynamic(self):
        return self.dynamic

    def SetDirLightAngles(self, zenith, azimuth):
        self.zenith, self.azimuth = float(zenith), float(azimuth)
        self.scene["dir_light_dir"] = dir_light_dir(self.zenith, self.azimuth)

    def Update(self, delta_time):
        """returns True when a scene parameter changed"""
        if not self.dynamic or self.scene_id != 3:
            return False
        self.SetDirLightAngles(self.zenith, math.fmod(self.azimuth + delta_time * 0.5, 2.0 * 3.141))
        return True


def frame_randoms(n, seed=1337):
    """Per-frame UniformData.random (src/NrcHpmRenderer.cu:308): n x 4 floats in [0,1), seeded (std::mt19937-like role)."""
    rng = np.random.default_rng(seed)
    return rng.random((n, 4), dtype=np.float32)
