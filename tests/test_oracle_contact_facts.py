"""Known-answer tests for the three MuJoCo contact facts the oracle takes from MuJoCo's documentation and that nothing in the reference's tests pins
(MuJoCo itself is not installable here, tests/test_physics_golden.py is the real pin):

  1. friction cone + impratio: a solid sphere on an incline rolls without slipping (a = 5/7 g sin) while the friction it needs, 2/7 m g sin, stays
     inside the cone, i.e. below tan(theta) = 3.5 mu, and slides at g (sin - mu cos) above it; while it rolls, the regularised friction rows let the
     contact point creep at a rate that shrinks with impratio (R_friction = R_normal / impratio);
  2. solref / solimp: a penetrating contact is the documented spring-damper  a = d(r) * (-B v - K d(r) r)  with K, B from (timeconst, dampratio)
     and the impedance d(r) from solimp -- checked against an independent numpy integration of that ODE;
  3. pyramidal vs elliptic regularisation: the four pyramid rows get R_py = 2 mu^2 R with R from diagApprox = tran (1 + mu^2); at constant impedance a
     resting body then sinks mu^2 (1 + mu^2) / 2 times as deep under the pyramidal cone as under the elliptic one, and both cones carry exactly its weight.

Each case is a single free sphere on the world plane (one contact, through the centre of mass -> the closed forms are exact).
"""

import copy

import numpy as np
import pytest

from oracle import oracle as O

G = 9.81


def _sphere_on_plane(cone="elliptic", impratio=1.0, mu=0.5, gravity=(0.0, 0.0, -G), dt=0.002, solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0), r=0.05, m=0.3):
    base = O.load_description("leap_cube")
    world, cube = copy.deepcopy(base["bodies"][0]), copy.deepcopy(next(b for b in base["bodies"] if b["name"] == "cube"))
    I = 0.4 * m * r * r
    cube.update(name="ball", parent=0, pos=[0.0, 0.0, r], mass=m, inertia=[I, I, I])
    jnt = copy.deepcopy(next(j for j in base["joints"] if j["type"] == "free"))
    jnt.update(name="ball_free", body=1)
    geom = dict(condim=3, friction=[mu, 0.005, 0.0001], solref=list(solref), solimp=list(solimp), margin=0.0, gap=0.0, solmix=1.0, priority=0, quat=[1.0, 0, 0, 0])
    plane = dict(geom, name="ground", body=0, type="plane", pos=[0.0, 0.0, 0.0], size=[10.0, 10.0, 0.01])
    ball = dict(geom, name="ball", body=1, type="sphere", pos=[0.0, 0.0, 0.0], size=[r])
    desc = dict(task="ball", source="tests", option=dict(timestep=dt, integrator="implicitfast", cone=cone, impratio=impratio, gravity=list(gravity), contact=True),
                bodies=[world, cube], joints=[jnt], geoms=[plane, ball], sites=[], actuators=[], sensors=[], excludes=[], equalities=[], nsensordata=0)
    return O.Model("ball", desc), r, m


def _impedance(si, dist):
    d0, d1, width, mid, power = si
    x = min(abs(dist) / width, 1.0)
    if power == 1:
        y = x
    elif x <= mid:
        y = x**power / mid ** (power - 1)
    else:
        y = 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    return d0 + y * (d1 - d0)


@pytest.mark.parametrize("cone", ["elliptic", "pyramidal"])
def test_incline_stick_slip_threshold_and_impratio_creep(cone):
    mu, H, dt = 0.2, 500, 0.002
    th_slip, th_stick = np.arctan(3.5 * mu) + 0.15, np.arctan(3.5 * mu) - 0.15

    def run(theta, impratio):
        m, r, _ = _sphere_on_plane(cone, impratio, mu, gravity=(G * np.sin(theta), 0.0, -G * np.cos(theta)), dt=dt)
        x0 = np.zeros(13)
        x0[2], x0[3] = r - 2e-4, 1.0
        st, _ = m.rollout(x0, np.zeros((1, H, 0)))
        return st[0]

    # above the threshold the contact point slides: the centre accelerates at g (sin - mu cos) (a sliding sphere with kinetic friction mu N picks up spin
    # as well; the linear acceleration is what the cone fixes)
    s = run(th_slip, 1.0)
    acc = (s[-1, 7] - s[H // 2, 7]) / ((H - 1 - H // 2) * dt)
    assert acc == pytest.approx(G * (np.sin(th_slip) - mu * np.cos(th_slip)), rel=0.03)
    # below it the friction rows hold: a rolling sphere (no slip at the contact) accelerates at (5/7) g sin(theta), contact point velocity ~ 0
    s1 = run(th_stick, 1.0)
    acc1 = (s1[-1, 7] - s1[H // 2, 7]) / ((H - 1 - H // 2) * dt)
    assert acc1 == pytest.approx(5.0 / 7.0 * G * np.sin(th_stick), rel=0.03)
    # velocity of the contact point (midway between the plane and the sphere's lowest point: lever arm (z + r) / 2; the ball rolls about y only, so the
    # body-frame omega_y is the world one): zero up to the creep the regularised friction rows allow, v_s = R_t a0 / (A_t B), which shrinks with impratio
    slip = lambda st: abs(st[-1, 7] - 0.5 * (st[-1, 2] + 0.05) * st[-1, 11])  # noqa: E731
    slip1, slip100 = slip(s1), slip(run(th_stick, 100.0))
    assert slip1 < 1e-3 * abs(s1[-1, 7]) and slip100 < 0.2 * slip1


def test_contact_spring_damper_follows_solref_and_solimp():
    solref, solimp, dt = (0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0), 0.001
    m, r, mass = _sphere_on_plane("elliptic", 1.0, 0.5, gravity=(0.0, 0.0, 0.0), dt=dt, solref=solref, solimp=solimp)
    pen0 = 3e-3  # deeper than the solimp width: the impedance starts at dmax and moves along the curve as the ball is pushed out
    x0 = np.zeros(13)
    x0[2], x0[3] = r - pen0, 1.0
    H = 60
    st, _ = m.rollout(x0, np.zeros((1, H, 0)))
    # documented model (MuJoCo computation chapter): K = 1 / (dmax^2 tc^2 dr^2), B = 2 / (dmax tc); aref = -B v - K d r; the single normal row has
    # A = 1/m and R = (1 - d) / d * A, so the constrained acceleration is d * aref while the contact is active
    tc, dr = solref
    K, B = 1.0 / (solimp[1] ** 2 * tc**2 * dr**2), 2.0 / (solimp[1] * tc)
    z, v = x0[2], 0.0
    for h in range(H):
        dist = z - r
        a = 0.0
        if dist < 0:
            d = _impedance(solimp, dist)
            a = max(0.0, d * (-B * v - K * d * dist))
        v += dt * a
        z += dt * v
        assert st[0, h, 2] == pytest.approx(z, abs=2e-9) and st[0, h, 9] == pytest.approx(v, abs=2e-6)
    assert st[0, -1, 9] > 0 and st[0, -1, 2] > x0[2]  # pushed out


def test_pyramidal_and_elliptic_cones_carry_the_weight_with_the_documented_regularisation():
    mu = 0.6
    pen, force = {}, {}
    for cone in ("elliptic", "pyramidal"):
        m, r, mass = _sphere_on_plane(cone, 1.0, mu, dt=0.002, solimp=(0.9, 0.9, 0.001, 0.5, 2.0))  # constant impedance: the closed form is exact
        x0 = np.zeros(13)
        x0[2], x0[3] = r, 1.0
        st, _ = m.rollout(x0, np.zeros((1, 2000, 0)))
        assert abs(st[0, -1, 9]) < 1e-6  # at rest
        pen[cone] = r - st[0, -1, 2]
        out = m.forward(st[0, -1, :7], st[0, -1, 7:], np.zeros(1))
        force[cone] = out["qfrc_constraint"][2]
        assert force[cone] == pytest.approx(mass * G, rel=1e-4)
    # elliptic: one normal row, m g = K d x / R0 with R0 = (1 - d) / (d m)  ->  x = g (1 - d) / (K d^2); pyramidal: four rows share the load, each with
    # R_py = 2 mu^2 (1 - d) / d * (1 + mu^2) / m  ->  x_py / x_el = R_py / (4 R0) = mu^2 (1 + mu^2) / 2
    d, K = 0.9, 1.0 / (0.9**2 * 0.02**2)
    assert pen["elliptic"] == pytest.approx(G * (1 - d) / (K * d * d), rel=1e-3)
    assert pen["pyramidal"] / pen["elliptic"] == pytest.approx(mu * mu * (1 + mu * mu) / 2, rel=1e-3)
