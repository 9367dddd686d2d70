"""Synthetic scene generator for bench.py and the full-size tests: a textured terrain patch
rendered through the reference's own camera models, so that every pixel has a known 3-D point.

  terrain   z(a, b): 5 octaves of value noise (lacunarity 2, gain 0.5, base wavelength 1/8 of the patch), +-2 km on a
            25 km x 25 km patch
  texture   finer octaves of value noise (wavelengths 32 .. 2 pixels of the nominal ground sampling distance, so that the
            feature density per pixel does not depend on the image size) + Gaussian blobs (sigma 1.5..6 px, amplitude
            +-40, one per 256 pixels), mean 128, quantised to u8 after a per-view sensor noise of 0.5 DN
  random    PCG32 (XSH-RR), seed 0x53524C43 for the scene, 0x53524C43 + 1 + view index for a view's sensor noise
  cameras   pinhole: patterned on the reference's fixtures (test/checkpoints/*/N_Image.cpimg): foc 0.8593, fov 0.04189 rad,
            dpix = foc tan(fov/2) / (W/2), first camera at the fixture's ECEF offset (402 km above a 6371 km sphere), the
            others on +-70 km baselines, all looking at the patch centre.  Pixel -> ray is generateBundle's model
            (src/PointCloudFactory.cu:4166-4199): ray = R(cam_rot) (dpix (x - W/2), dpix (y - H/2), foc).
            pushbroom: generatePushbroomBundle's model (src/PointCloudFactory.cu:4201-4283), one roll per view.

Everything heavy runs in torch on the current device (CUDA on the GPU box, CPU for the small sizes of CPU tests).
This is input synthesis for measurements and tests, not part of the product path.
"""
import math

import numpy as np
import torch

SEED = 0x53524C43
PATCH_KM = 25.0
TERRAIN_AMPLITUDE_KM = 2.0
FIXTURE_ECEF = np.array([321.2984924316406, 5980.93310546875, 3163.05615234375])  # camera 0 of the reference's fixtures
FIXTURE_FOC, FIXTURE_FOV = 0.8593109846115112, 0.04188790172338486
EARTH_RADIUS_KM = 6371.0
# texture octaves below the terrain's: value noise at these wavelengths (image pixels), amplitude 30 DN falling by 0.7 per
# octave -- fine detail a few DN strong, so that the DoG extrema come from the scene (and repeat across views) rather
# than from the 0.5 DN sensor noise
TEXTURE_WAVELENGTHS_PX = (32, 16, 8, 4, 2)
TEXTURE_AMPLITUDE, TEXTURE_GAIN = 30.0, 0.7
SENSOR_NOISE_DN = 0.5

CAMERA = np.dtype({"names": ["cam_pos", "cam_rot", "fov", "foc", "dpix", "timeStamp", "ecef_offset", "no_rot", "size"],
                   "formats": [("<f4", (3,)), ("<f4", (3,)), ("<f4", (2,)), "<f4", ("<f4", (2,)), "<i8", ("<f4", (3,)), "u1",
                               ("<u4", (2,))],
                   "offsets": [0, 12, 24, 32, 40, 48, 56, 68, 72], "itemsize": 80})
PUSHBROOM = np.dtype([("start_pos", "<f4", (3,)), ("end_pos", "<f4", (3,)), ("projection_center", "<f4", (2,)),
                      ("axis_radius", "<f4"), ("roll", "<f4"), ("altitude", "<f4"), ("foc", "<f4"), ("fov", "<f4"),
                      ("gsd", "<f4"), ("dpix", "<f4", (2,)), ("size", "<u4", (2,))])


# ---------------------------------------------------------------------------------------------------------------- PCG32
def pcg32(n, seed, stream=0):
    """n outputs of PCG32 (XSH-RR 64/32), vectorised: the LCG state after k steps is a^k s0 + c (a^k - 1)/(a - 1)."""
    a = np.uint64(6364136223846793005)
    inc = np.uint64((int(stream) << 1 | 1) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        state0 = (np.uint64(0) * a + inc)
        state0 = (state0 + np.uint64(seed)) * a + inc                  # pcg32_srandom_r
        ak = np.cumprod(np.concatenate([[np.uint64(1)], np.full(n, a, np.uint64)]))       # a^0 .. a^n
        geo = np.concatenate([[np.uint64(0)], np.cumsum(ak[:-1], dtype=np.uint64)])       # sum_{j<k} a^j
        states = ak * state0 + inc * geo                                                   # state before output k
        old = states[:n]
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        out = (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))
    return out.astype(np.uint32)


def pcg_uniform(n, seed, stream=0):
    return pcg32(n, seed, stream).astype(np.float64) / 4294967296.0


# ---------------------------------------------------------------------------------------------------------------- noise
def _value_noise(coords_a, coords_b, cells, lattice):
    """Smoothstep-interpolated lattice noise in [-1, 1]; coords in [0, 1) patch units, lattice (cells+1)^2 torch values."""
    fa, fb = coords_a * cells, coords_b * cells
    ia = torch.clamp(fa.floor().long(), 0, cells - 1)
    ib = torch.clamp(fb.floor().long(), 0, cells - 1)
    ta, tb = fa - ia, fb - ib
    ta = ta * ta * (3 - 2 * ta)
    tb = tb * tb * (3 - 2 * tb)
    n = cells + 1
    v00, v10 = lattice[ib * n + ia], lattice[ib * n + ia + 1]
    v01, v11 = lattice[(ib + 1) * n + ia], lattice[(ib + 1) * n + ia + 1]
    return (v00 * (1 - ta) + v10 * ta) * (1 - tb) + (v01 * (1 - ta) + v11 * ta) * tb


def _blur_separable(img, g, r):
    """Zero-padded separable convolution as 2 (2r + 1) shifted multiply-adds (no conv library: MIOpen spends minutes
    choosing a kernel for a 20480^2 single-channel image)."""
    n = img.shape[0]
    pad = torch.zeros(n, n + 2 * r, device=img.device)
    pad[:, r: r + n] = img
    out = torch.zeros_like(img)
    for k in range(2 * r + 1):
        out += g[k] * pad[:, k: k + n]
    pad = torch.zeros(n + 2 * r, n, device=img.device)
    pad[r: r + n] = out
    out = torch.zeros_like(img)
    for k in range(2 * r + 1):
        out += g[k] * pad[k: k + n]
    return out


class Scene:
    """Terrain + texture of one patch.  `texels` = side of the texture raster (2 per nominal image pixel)."""

    def __init__(self, image_size, gsd_km, device=None, seed=SEED, anisotropy=1.0):
        self.device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        self.gsd = float(gsd_km)
        self.aniso = float(anisotropy)      # ground size of a pixel along b relative to a (pushbroom strips: << 1)
        dev = self.device
        # terrain lattices: 8, 16, 32, 64, 128 cells
        self.terrain = []
        stream = 1
        for o in range(5):
            cells = 8 << o
            vals = pcg_uniform((cells + 1) ** 2, seed, stream) * 2 - 1
            self.terrain.append((cells, torch.from_numpy(vals).to(dev, torch.float32), 0.5 ** o))
            stream += 1
        self.terrain_norm = sum(g for _, _, g in self.terrain)
        # texture raster: covers image_size * 1.25 pixels at half-pixel texels, centred on the patch centre
        self.tex_n = int(image_size * 2.5) // 8 * 8
        self.texel_a = self.gsd / 2.0
        self.texel_b = self.gsd * self.aniso / 2.0
        n = self.tex_n
        ramp = (torch.arange(n, device=dev, dtype=torch.float32) + 0.5) / n
        tb, ta = torch.meshgrid(ramp, ramp, indexing="ij")
        tex = torch.zeros(n, n, device=dev)
        for k, wl in enumerate(TEXTURE_WAVELENGTHS_PX):  # wavelengths in image pixels = 2 * wl texels
            cells = max(n // (2 * wl), 1)
            vals = pcg_uniform((cells + 1) ** 2, seed, stream) * 2 - 1
            stream += 1
            tex += TEXTURE_AMPLITUDE * (TEXTURE_GAIN ** k) * _value_noise(ta, tb, cells, torch.from_numpy(vals).to(dev, torch.float32))
        del ta, tb
        # blobs: one per 256 image pixels, four sigma classes (image pixels), splat as impulses then blurred
        nblob = max((image_size * image_size) // 256, 16)
        u = pcg_uniform(nblob * 4, seed, stream).reshape(4, nblob)
        stream += 1
        cls = (u[2] * 4).astype(np.int64)
        amp = (u[3] * 2 - 1) * 40.0
        pa = torch.from_numpy((u[0] * (n - 1)).astype(np.int64)).to(dev)
        pb = torch.from_numpy((u[1] * (n - 1)).astype(np.int64)).to(dev)
        for c, sig in enumerate((1.5, 2.5, 4.0, 6.0)):
            sel = torch.from_numpy(cls == c).to(dev)
            imp = torch.zeros(n * n, device=dev)
            # impulse weight = amplitude * (2 pi sigma_t^2) so that the blurred blob peaks at `amplitude`
            st = 2.0 * sig
            imp.index_add_(0, (pb[sel] * n + pa[sel]), torch.from_numpy(amp[cls == c]).to(dev, torch.float32) * (2 * math.pi * st * st))
            r = int(math.ceil(3 * st))
            g = torch.exp(-0.5 * (torch.arange(-r, r + 1, device=dev, dtype=torch.float32) / st) ** 2)
            g = g / g.sum()
            tex += _blur_separable(imp.view(n, n), g, r)
        self.tex = (tex + 128.0).contiguous()

    def height(self, a, b):
        """Terrain height (km) at patch coordinates a, b (km, centre = 0)."""
        ua = torch.clamp(a / PATCH_KM + 0.5, 0.0, 0.999999)
        ub = torch.clamp(b / PATCH_KM + 0.5, 0.0, 0.999999)    # the terrain is isotropic in km; only the texture follows the pixel grid
        h = torch.zeros_like(a)
        for cells, lat, gain in self.terrain:
            h = h + gain * _value_noise(ua, ub, cells, lat)
        return h * (TERRAIN_AMPLITUDE_KM / self.terrain_norm)

    def texture(self, a, b):
        """Bilinear texture sample at patch coordinates (km)."""
        n = self.tex_n
        ga = a / (self.texel_a * n / 2.0)
        gb = b / (self.texel_b * n / 2.0)
        grid = torch.stack([ga, gb], -1).view(1, 1, -1, 2)
        out = torch.nn.functional.grid_sample(self.tex.view(1, 1, n, n), grid, mode="bilinear", padding_mode="border",
                                              align_corners=False)
        return out.view(a.shape)


# ---------------------------------------------------------------------------------------------------------------- pinhole
def _rotation_to_euler(M):
    """rotatePoint's matrix is Rz(z) Ry(y) Rx(x) (src/matrix_util.cu:314-327): angles of a rotation matrix M."""
    y = -math.asin(max(-1.0, min(1.0, M[2, 0])))
    x = math.atan2(M[2, 1], M[2, 2])
    z = math.atan2(M[1, 0], M[0, 0])
    return np.array([x, y, z])


def _euler_matrix(rot):
    x, y, z = [float(v) for v in rot]
    cx, sx, cy, sy, cz, sz = math.cos(x), math.sin(x), math.cos(y), math.sin(y), math.cos(z), math.sin(z)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


class PinholeRig:
    """V cameras around the fixture geometry, all looking at the patch centre."""

    def __init__(self, num_views, size, seed=SEED):
        self.size = int(size)
        O = FIXTURE_ECEF
        down = -O / np.linalg.norm(O)
        self.centre = down * (np.linalg.norm(O) - EARTH_RADIUS_KM)          # patch centre, relative to camera 0
        e1 = np.cross(down, [0.0, 0.0, 1.0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(down, e1)
        self.e1, self.e2, self.down = e1, e2, down
        self.gsd = 2.0 * np.linalg.norm(self.centre) * math.tan(FIXTURE_FOV / 2) / self.size
        jit = pcg_uniform(num_views * 3, seed, 99).reshape(num_views, 3) * 2 - 1
        cams = np.zeros(num_views, CAMERA)
        self.M = []
        for v in range(num_views):
            # baselines: 0, -70, +70, -35, +35, ... km along e1, a few km of jitter along e2 / down
            base = 0.0 if v == 0 else (70.0 if v % 2 == 0 else -70.0) / (1 + (v - 1) // 2)
            pos = base * e1 + (jit[v, 1] * 3.0) * e2 - (jit[v, 2] * 1.5) * down if v else np.zeros(3)
            a3 = self.centre - pos
            a3 /= np.linalg.norm(a3)
            a1 = e1 - np.dot(e1, a3) * a3
            a1 /= np.linalg.norm(a1)
            a2 = np.cross(a3, a1)
            M = np.stack([a1, a2, a3], 1)
            rot = _rotation_to_euler(M).astype(np.float32)
            cams["cam_pos"][v] = pos
            cams["cam_rot"][v] = rot
            cams["fov"][v] = FIXTURE_FOV
            cams["foc"][v] = FIXTURE_FOC
            dp = np.float32(np.float32(FIXTURE_FOC) * np.float32(math.tan(np.float32(FIXTURE_FOV) / 2.0))) / np.float32(self.size / 2.0)
            cams["dpix"][v] = dp
            cams["ecef_offset"][v] = O
            cams["size"][v] = self.size
            self.M.append(_euler_matrix(cams["cam_rot"][v]))               # the matrix the float32 angles really give
        self.cameras = cams

    def ground_points(self, scene, view, xs, ys):
        """3-D points (km, in the cameras' frame) seen at pixel coordinates xs, ys of `view`: ray / terrain intersection."""
        dev = scene.device
        cam = self.cameras[view]
        dp, foc = float(cam["dpix"][0]), float(cam["foc"])
        M = torch.tensor(self.M[view], device=dev, dtype=torch.float64)
        half = self.size / 2.0
        d = torch.stack([(xs.double() - half) * dp, (ys.double() - half) * dp, torch.full_like(xs, foc, dtype=torch.float64)], -1) @ M.T
        d = d / d.norm(dim=-1, keepdim=True)
        p0 = torch.tensor(cam["cam_pos"].astype(np.float64) - self.centre, device=dev)
        e1 = torch.tensor(self.e1, device=dev)
        e2 = torch.tensor(self.e2, device=dev)
        dn = torch.tensor(self.down, device=dev)
        p0d, dd = (p0 * dn).sum(), (d * dn).sum(-1)
        t = -p0d / dd                                                        # plane through the patch centre
        for _ in range(6):                                                   # point.down = -height(a, b)
            q = p0 + t.unsqueeze(-1) * d
            h = scene.height((q * e1).sum(-1).float(), (q * e2).sum(-1).float()).double()
            t = (-h - p0d) / dd
        q = p0 + t.unsqueeze(-1) * d
        return q + torch.tensor(self.centre, device=dev), (q * e1).sum(-1).float(), (q * e2).sum(-1).float()

    def render(self, scene, view, seed=SEED):
        """u8 image (H, W) of `view` on scene.device."""
        n = self.size
        dev = scene.device
        out = torch.empty(n, n, dtype=torch.uint8, device=dev)
        rows = max(1, (1 << 22) // n)
        noise_seed = seed + 1 + view
        for y0 in range(0, n, rows):
            y1 = min(n, y0 + rows)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, device=dev, dtype=torch.float32),
                                    torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
            _, a, b = self.ground_points(scene, view, xs.reshape(-1), ys.reshape(-1))
            val = scene.texture(a, b)
            u = torch.from_numpy(pcg_uniform(2 * val.numel(), noise_seed, y0 + 1).astype(np.float32)).to(dev)
            gauss = torch.sqrt(-2 * torch.log(u[0::2].clamp_min(1e-12))) * torch.cos(2 * math.pi * u[1::2])
            out[y0:y1] = torch.clamp(torch.round(val + SENSOR_NOISE_DN * gauss), 0, 255).to(torch.uint8).view(y1 - y0, n)
        return out


def pinhole_views(num_views, size, device=None, seed=SEED):
    """-> (list of u8 (size, size) tensors, Image::Camera array, rig, scene)"""
    rig = PinholeRig(num_views, size, seed)
    scene = Scene(size, rig.gsd, device, seed)
    return [rig.render(scene, v, seed) for v in range(num_views)], rig.cameras, rig, scene


# -------------------------------------------------------------------------------------------------------------- pushbroom
class PushbroomRig:
    """V pushbroom strips of the same ground with different rolls; pixel -> line is generatePushbroomBundle's model
    (src/PointCloudFactory.cu:4201-4283): the craft sits where the line of slope tan(roll - pi/2) through the
    projection centre meets the orbit circle, image line y turns it about the x axis by gsd (y - cy) / radius, the look
    vector is (dpix (x - cx), 0, -foc) rolled about y."""

    def __init__(self, num_views, size, seed=SEED, gsd_km=0.006, altitude=400.0, radius=EARTH_RADIUS_KM):
        self.size = int(size)
        pb = np.zeros(num_views, PUSHBROOM)
        # positive rolls only: for roll < 0 the model's root selection (`solution1 > 0`) picks the far side of the planet,
        # and roll = 0 is a pole of tan(roll - pi/2)
        rolls = np.linspace(2.0, 16.0, num_views) if num_views > 1 else np.array([8.0])
        pb["axis_radius"], pb["altitude"], pb["roll"] = radius, altitude, rolls
        pb["foc"], pb["fov"], pb["gsd"] = FIXTURE_FOC, FIXTURE_FOV, gsd_km
        pb["dpix"][:, 0] = np.float32(np.float32(FIXTURE_FOC) * np.float32(math.tan(np.float32(FIXTURE_FOV) / 2.0))) / np.float32(size / 2.0)
        pb["size"] = size
        self.cameras = pb
        self.altitude, self.radius = altitude, radius
        self.gsd_x = 2.0 * altitude * math.tan(FIXTURE_FOV / 2) / size            # ground size of a pixel across track
        self.gsd_y = gsd_km * altitude / radius                                    # ... and along track, as the model has it

    def lines(self, view, xs, ys):
        """(pnt, vec) of generatePushbroomBundle in float64 torch."""
        pb = self.cameras[view]
        half = self.size / 2.0
        roll = float(pb["roll"]) * math.pi / 180.0
        t = math.tan(roll - math.pi / 2)
        radius, alt = float(pb["axis_radius"]), float(pb["altitude"])
        a, b, c = 1 + t * t, -2 * radius * t, radius * radius - (alt + radius) ** 2
        s1 = (-b + math.sqrt(b * b - 4 * a * c)) / (2 * a)
        s2 = (-b - math.sqrt(b * b - 4 * a * c)) / (2 * a)
        sol = s1 if s1 > 0 else s2
        pos = np.array([sol, 0.0, -t * sol])
        ang = (ys.double() - half) * (float(pb["gsd"]) / radius)
        ca, sa = torch.cos(ang), torch.sin(ang)
        # rotatePoint(position, {angle_out, 0, 0}): rotation about x
        pnt = torch.stack([torch.full_like(ang, pos[0]), pos[1] * ca - pos[2] * sa, pos[1] * sa + pos[2] * ca], -1)
        kx = float(pb["dpix"][0]) * (xs.double() - half)
        kz = -float(pb["foc"])
        cr, sr = math.cos(roll), math.sin(roll)
        # rotatePoint(k, {0, roll, 0}): rotation about y
        vec = torch.stack([kx * cr + kz * sr, torch.zeros_like(kx), -kx * sr + kz * cr], -1)
        return pnt, vec / vec.norm(dim=-1, keepdim=True)

    def ground_points(self, scene, view, xs, ys):
        pnt, vec = self.lines(view, xs, ys)
        t = -pnt[:, 2] / vec[:, 2]
        for _ in range(6):
            q = pnt + t.unsqueeze(-1) * vec
            h = scene.height(q[:, 0].float(), q[:, 1].float()).double()
            t = (h - pnt[:, 2]) / vec[:, 2]
        q = pnt + t.unsqueeze(-1) * vec
        return q, q[:, 0].float(), q[:, 1].float()

    def render(self, scene, view, seed=SEED):
        n = self.size
        dev = scene.device
        out = torch.empty(n, n, dtype=torch.uint8, device=dev)
        rows = max(1, (1 << 22) // n)
        for y0 in range(0, n, rows):
            y1 = min(n, y0 + rows)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, device=dev, dtype=torch.float32),
                                    torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
            _, a, b = self.ground_points(scene, view, xs.reshape(-1), ys.reshape(-1))
            val = scene.texture(a, b)
            u = torch.from_numpy(pcg_uniform(2 * val.numel(), seed + 1 + view, y0 + 1).astype(np.float32)).to(dev)
            gauss = torch.sqrt(-2 * torch.log(u[0::2].clamp_min(1e-12))) * torch.cos(2 * math.pi * u[1::2])
            out[y0:y1] = torch.clamp(torch.round(val + SENSOR_NOISE_DN * gauss), 0, 255).to(torch.uint8).view(y1 - y0, n)
        return out


def pushbroom_views(num_views, size, device=None, seed=SEED):
    """-> (list of u8 (size, size) strips, PushbroomCamera array, rig, scene)"""
    rig = PushbroomRig(num_views, size, seed)
    scene = Scene(size, rig.gsd_x, device, seed, anisotropy=rig.gsd_y / rig.gsd_x)
    return [rig.render(scene, v, seed) for v in range(num_views)], rig.cameras, rig, scene
