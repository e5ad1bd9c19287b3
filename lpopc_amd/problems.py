"""Problem builders for the BASELINE.json configs, written against the mirrored set-up API
(lpopc_amd.problem) exactly the way the reference's example main()s are written.

  launch()          — Delta-III 4-phase ascent, example/launch/Launch.cpp:76-547 (metric config)
  hypersensitive()  — example/hypersensitive/HyperSensitive.cpp:15-71
  bryson_denham()   — example/bryson-denham/BrysonDenham.cpp:9-98
  brachistochrone(), min_time_climb(), quadrotor() — authored here (not in the reference)

plus the synthetic meshes and seeded NLP iterates SURVEY.md §8(d) prescribes.
"""
import math

import numpy as np

from .problem import Linkage, OptimalProblem, Phase, ProblemFunctor

RPM_PROBLEM_LAUNCH = 1
RPM_PROBLEM_HYPERSENSITIVE = 2
RPM_PROBLEM_BRYSON_DENHAM = 3
RPM_PROBLEM_BRACHISTOCHRONE = 4
RPM_PROBLEM_MIN_TIME_CLIMB = 5
RPM_PROBLEM_QUADROTOR = 6
RPM_PROBLEM_PARAM_SLED = 7
RPM_PROBLEM_PARAM_OSC = 8


def set_uniform_mesh(phase, n_intervals, nodes):
    if n_intervals is None:
        return
    for i in range(n_intervals + 1):
        phase.SetMeshPoints(-1.0 + 2.0 * i / n_intervals if i < n_intervals else 1.0)
    for _ in range(n_intervals):
        phase.SetNodesPerInterval(nodes)


def set_mesh(phase, mesh_points, nodes_per_interval):
    for m in mesh_points:
        phase.SetMeshPoints(m)
    for n in nodes_per_interval:
        phase.SetNodesPerInterval(n)


# --------------------------------------------------------------------------- Delta-III
def _launch_oe2rv(oe, mu):
    """Launchoe2rv, example/launch/Launch.cpp:549-590"""
    a, e, i, Om, om, nu = oe
    p = a * (1 - e * e)
    r = p / (1 + e * math.cos(nu))
    rv = np.array([r * math.cos(nu), r * math.sin(nu), 0.0])
    vv = np.array([-math.sin(nu), e + math.cos(nu), 0.0]) * math.sqrt(mu / p)
    cO, sO, co, so, ci, si = math.cos(Om), math.sin(Om), math.cos(om), math.sin(om), math.cos(i), math.sin(i)
    R = np.array([[cO * co - sO * so * ci, -cO * so - sO * co * ci, sO * si],
                  [sO * co + cO * so * ci, -sO * so + cO * co * ci, -cO * si],
                  [so * si, co * si, ci]])
    return R @ rv, R @ vv


def launch(n_intervals=None, nodes=None):
    """Delta-III launch vehicle ascent.  n_intervals/nodes=None keeps the reference's default
    mesh (1 interval x 20 nodes per phase); the metric config is launch(64, 16)."""
    PI = math.pi
    earthRadius = 6378145.0
    gravParam = 3.986012e14
    initialMass = 301454.0
    earthRotRate = 7.29211585e-5
    seaLevelDensity = 1.225
    densityScaleHeight = 7200.0
    g0 = 9.80665
    # struct_scales, Launch.cpp:24-45
    s_length = earthRadius
    s_speed = math.sqrt(gravParam / s_length)
    s_time = s_length / s_speed
    s_acc = s_speed / s_time
    s_mass = initialMass
    s_force = s_mass * s_acc
    s_area = s_length * s_length
    s_volume = s_area * s_length
    s_density = s_mass / s_volume
    s_gravparam = s_acc * s_length * s_length
    omega = earthRotRate * s_time
    C_mu = gravParam / s_gravparam
    C_cd = 0.5
    C_sa = 4 * PI / s_area
    C_rho0 = seaLevelDensity / s_density
    C_H = densityScaleHeight / s_length
    C_Re = earthRadius / s_length
    C_g0 = g0 / s_acc
    omega_matrix = [0, 1 * omega, 0, -1 * omega, 0, 0, 0, 0, 0]  # 8 initialisers + implicit 0 (SURVEY B-17)

    lat0 = 28.5 * PI / 180
    x0 = C_Re * math.cos(lat0)
    z0 = C_Re * math.sin(lat0)
    y0 = 0.0
    r0 = np.array([x0, y0, z0])
    om = np.array(omega_matrix).reshape(3, 3, order="F")
    v0 = om @ r0

    bt_srb = 75.2 / s_time
    bt_first = 261.0 / s_time
    bt_second = 700.0 / s_time
    t0 = 0.0 / s_time
    t1 = 75.2 / s_time
    t2 = 150.4 / s_time
    t3 = 261 / s_time
    t4 = 961 / s_time
    m_tot_srb = 19290 / s_mass
    m_prop_srb = 17010 / s_mass
    m_dry_srb = m_tot_srb - m_prop_srb
    m_tot_first = 104380 / s_mass
    m_prop_first = 95550 / s_mass
    m_dry_first = m_tot_first - m_prop_first
    m_tot_second = 19300 / s_mass
    m_prop_second = 16820 / s_mass
    m_payload = 4164 / s_mass
    thrust_srb = 628500 / s_force
    thrust_first = 1083100 / s_force
    thrust_second = 110094 / s_force
    mdot_srb = m_prop_srb / bt_srb
    ISP_srb = thrust_srb / (C_g0 * mdot_srb)
    mdot_first = m_prop_first / bt_first
    ISP_first = thrust_first / (C_g0 * mdot_first)
    mdot_second = m_prop_second / bt_second
    ISP_second = thrust_second / (C_g0 * mdot_second)

    af = 24361140 / s_length
    ef = 0.7308
    incf = 28.5 * PI / 180
    Omf = 269.8 * PI / 180
    omf = 130.5 * PI / 180
    nuguess = 0.0
    rout, vout = _launch_oe2rv([af, ef, incf, Omf, omf, nuguess], C_mu)
    m10 = m_payload + m_tot_second + m_tot_first + 9 * m_tot_srb
    m1f = m10 - (6 * mdot_srb + mdot_first) * t1
    m20 = m1f - 6 * m_dry_srb
    m2f = m20 - (3 * mdot_srb + mdot_first) * (t2 - t1)
    m30 = m2f - 3 * m_dry_srb
    m3f = m30 - mdot_first * (t3 - t2)
    m40 = m3f - m_dry_first
    m4f = m_payload

    consts = omega_matrix + [C_mu, C_cd, C_sa, C_rho0, C_H, C_Re, C_g0, thrust_srb, thrust_first,
                             thrust_second, ISP_srb, ISP_first, ISP_second]
    rmin = -2 * C_Re
    rmax = -rmin
    vmin = -10000 / s_speed
    vmax = -vmin

    def common(ph, rguess, vguess, mg0, mgf):
        for _ in range(3):
            ph.SetcontrolMin(-1)
            ph.SetcontrolMax(1)
        ph.SetpathMin(1)
        ph.SetpathMax(1)
        for i in range(3):
            ph.SetStateGuess(i + 1, rguess[i])
            ph.SetStateGuess(i + 1, rguess[i])
        for i in range(3):
            ph.SetStateGuess(i + 4, vguess[i])
            ph.SetStateGuess(i + 4, vguess[i])
        ph.SetStateGuess(7, mg0)
        ph.SetStateGuess(7, mgf)
        for i, u in enumerate((0, 1, 0)):
            ph.SetControlGuess(i + 1, u)
            ph.SetControlGuess(i + 1, u)

    def free_rv_bounds(ph):
        for _ in range(3):
            ph.SetStateMin(rmin, rmin, rmin)
            ph.SetStateMax(rmax, rmax, rmax)
        for _ in range(3):
            ph.SetStateMin(vmin, vmin, vmin)
            ph.SetStateMax(vmax, vmax, vmax)

    P1 = Phase(1, 7, 3, 0, 1, 0)
    P1.SetTimeMin(t0, t1)
    P1.SetTimeMax(t0, t1)
    for i in range(3):
        P1.SetStateMin(r0[i], rmin, rmin)
        P1.SetStateMax(r0[i], rmax, rmax)
    for i in range(3):
        P1.SetStateMin(v0[i], vmin, vmin)
        P1.SetStateMax(v0[i], vmax, vmax)
    P1.SetStateMin(m10, m1f, m1f)
    P1.SetStateMax(m10, m10, m10)
    P1.SetTimeGuess(t0)
    P1.SetTimeGuess(t1)
    common(P1, r0, v0, m10, m1f)

    P2 = Phase(2, 7, 3, 0, 1, 0)
    P2.SetTimeMin(t1, t2)
    P2.SetTimeMax(t1, t2)
    free_rv_bounds(P2)
    P2.SetStateMin(m2f, m2f, m2f)
    P2.SetStateMax(m20, m20, m20)
    P2.SetTimeGuess(t1)
    P2.SetTimeGuess(t2)
    common(P2, r0, v0, m20, m2f)

    P3 = Phase(3, 7, 3, 0, 1, 0)
    P3.SetTimeMin(t2, t3)
    P3.SetTimeMax(t2, t3)
    free_rv_bounds(P3)
    P3.SetStateMin(m3f, m3f, m3f)
    P3.SetStateMax(m30, m30, m30)
    P3.SetTimeGuess(t2)
    P3.SetTimeGuess(t3)
    common(P3, rout, vout, m30, m3f)

    P4 = Phase(4, 7, 3, 0, 1, 5)
    P4.SetTimeMin(t3, t3)
    P4.SetTimeMax(t3, t4)
    free_rv_bounds(P4)
    P4.SetStateMin(m4f, m4f, m4f)
    P4.SetStateMax(m40, m40, m40)
    for v in (af, ef, incf, Omf, omf):
        P4.SeteventMin(v)
    for v in (af, ef, incf, Omf, omf):
        P4.SeteventMax(v)
    P4.SetTimeGuess(t3)
    P4.SetTimeGuess(t4)
    common(P4, rout, vout, m40, m4f)

    links = []
    for ipair, (l, r, dm) in enumerate(((1, 2, -6 * m_dry_srb), (2, 3, -3 * m_dry_srb), (3, 4, -m_dry_first))):
        lk = Linkage(ipair + 1, l, r)
        for _ in range(6):
            lk.SetLinkMin(0)
        lk.SetLinkMin(dm)
        for _ in range(6):
            lk.SetLinkMax(0)
        lk.SetLinkMax(dm)
        links.append(lk)

    op = OptimalProblem(4, 3, ProblemFunctor(RPM_PROBLEM_LAUNCH, consts))
    for ph in (P1, P2, P3, P4):
        set_uniform_mesh(ph, n_intervals, nodes)
        op.AddPhase(ph)
    for lk in links:
        op.AddLinkage(lk)
    return op


# --------------------------------------------------------------------------- Hypersensitive
def hypersensitive(mesh_points=None, nodes_per_interval=None, tf=5000.0):
    t0, x0, xf = 0.0, 1.5, 1.0
    xmin, xmax, umin, umax = -10, 10, -10, 10
    P1 = Phase(1, 1, 1, 0, 0, 0)
    P1.SetTimeMin(t0, tf)
    P1.SetTimeMax(t0, tf)
    P1.SetStateMin(x0, xmin, xf)
    P1.SetStateMax(x0, xmax, xf)
    P1.SetcontrolMin(umin)
    P1.SetcontrolMax(umax)
    P1.SetTimeGuess(t0)
    P1.SetTimeGuess(tf)
    P1.SetStateGuess(1, x0)
    P1.SetStateGuess(1, xf)
    P1.SetControlGuess(1, -1)
    P1.SetControlGuess(1, 1)
    if mesh_points is not None:
        set_mesh(P1, mesh_points, nodes_per_interval)
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_HYPERSENSITIVE, []))
    op.AddPhase(P1)
    return op


def hp_mesh(total_nodes=4096, ratio=1.05, seed=4):
    """Synthetic hp mesh of SURVEY §8(d) config 4: interval widths graded geometrically towards both
    ends of [-1,1] (boundary layers), N_k drawn from {4,8,12,16} by a seeded pattern with 16 in the
    layers and small orders in the flat middle, trimmed so that sum(N_k) == total_nodes."""
    rng = np.random.RandomState(seed)
    nodes = []
    K = 340
    half = K // 2
    for k in range(K):
        d = min(k, K - 1 - k) / float(half)  # 0 at the ends, 1 in the middle
        if d < 0.25:
            nodes.append(16)
        elif d < 0.5:
            nodes.append(int(rng.choice([12, 16])))
        elif d < 0.75:
            nodes.append(int(rng.choice([8, 12])))
        else:
            nodes.append(int(rng.choice([4, 8])))
    # trim / grow to the requested total in steps of 4, keeping every N_k in [4,16]
    k = 0
    while sum(nodes) != total_nodes:
        i = (k * 7919) % K
        if sum(nodes) > total_nodes and nodes[i] > 4:
            nodes[i] -= 4 if sum(nodes) - total_nodes >= 4 else sum(nodes) - total_nodes
            nodes[i] = max(nodes[i], 2)
        elif sum(nodes) < total_nodes and nodes[i] < 16:
            nodes[i] += min(4, total_nodes - sum(nodes))
        k += 1
    widths = np.array([ratio ** min(k, K - 1 - k) for k in range(K)], dtype=np.float64)
    edges = np.concatenate([[0.0], np.cumsum(widths)])
    mesh = -1.0 + 2.0 * edges / edges[-1]
    mesh[0], mesh[-1] = -1.0, 1.0
    return [float(m) for m in mesh], [int(n) for n in nodes]


# --------------------------------------------------------------------------- Bryson-Denham
def bryson_denham(n_intervals=None, nodes=None):
    P1 = Phase(1, 3, 1, 0, 0, 5)
    P1.SetTimeMin(0.0, 0.0)
    P1.SetTimeMax(0, 50)
    P1.SetStateMin(0, 0, 0)
    P1.SetStateMax(1.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0)
    P1.SetStateMin(-10, -10, -10)
    P1.SetStateMax(10, 10, 10)
    P1.SetStateMin(-10, -10, -10)
    P1.SetStateMax(10, 10, 10)
    P1.SetcontrolMin(-10)
    P1.SetcontrolMax(10)
    P1.SetTimeGuess(0.0)
    P1.SetTimeGuess(1.0)
    for v in (0, 1, 0, 0, -1):
        P1.SeteventMin(v)
    for v in (0, 1, 0, 0, -1):
        P1.SeteventMax(v)
    P1.SetStateGuess(1, 0)
    P1.SetStateGuess(1, 0)
    P1.SetStateGuess(2, 1.0)
    P1.SetStateGuess(2, -1.0)
    P1.SetStateGuess(3, 0.0)
    P1.SetStateGuess(3, 0.0)
    P1.SetControlGuess(1, 0.0)
    P1.SetControlGuess(1, 0.0)
    set_uniform_mesh(P1, n_intervals, nodes)
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_BRYSON_DENHAM, []))
    op.AddPhase(P1)
    return op


# --------------------------------------------------------------------------- authored problems
def brachistochrone(n_intervals=1, nodes=10):
    """BASELINE config 1: nx=3 (x,y,v), nu=1 (theta), ne=5 (x0,y0,v0,xf,yf); cost = tf."""
    g = 9.80665
    P1 = Phase(1, 3, 1, 0, 0, 5)
    P1.SetTimeMin(0.0, 0.0)
    P1.SetTimeMax(0.0, 10.0)
    for _ in range(2):
        P1.SetStateMin(0, 0, 0)
        P1.SetStateMax(10, 10, 10)
    P1.SetStateMin(0, 0, 0)
    P1.SetStateMax(20, 20, 20)
    P1.SetcontrolMin(0.0)
    P1.SetcontrolMax(math.pi)
    for v in (0, 0, 0, 2, 2):
        P1.SeteventMin(v)
    for v in (0, 0, 0, 2, 2):
        P1.SeteventMax(v)
    P1.SetTimeGuess(0.0)
    P1.SetTimeGuess(1.0)
    P1.SetStateGuess(1, 0.0)
    P1.SetStateGuess(1, 2.0)
    P1.SetStateGuess(2, 0.0)
    P1.SetStateGuess(2, 2.0)
    P1.SetStateGuess(3, 0.0)
    P1.SetStateGuess(3, 6.0)
    P1.SetControlGuess(1, 0.2)
    P1.SetControlGuess(1, 1.2)
    set_uniform_mesh(P1, n_intervals, nodes)
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_BRACHISTOCHRONE, [g]))
    op.AddPhase(P1)
    return op


def min_time_climb(n_intervals=16, nodes=16):
    """BASELINE config 2: nx=4 (h,v,gamma,m), nu=1 (alpha), ne=7; cost = tf.  Smooth-atmosphere
    variant authored here (model in DESIGN.md §Problems)."""
    Re, mu, S, g0, Isp = 6378145.0, 3.986e14, 49.2386, 9.80665, 1600.0
    rho0, Hs, a0, a1, Tmax = 1.225, 7254.24, 340.3, 0.0041, 1.2e5
    h0, v0, gam0, m0 = 0.0, 129.314, 0.0, 19050.864
    hf, vf, gamf = 19994.88, 295.092, 0.0
    P1 = Phase(1, 4, 1, 0, 0, 7)
    P1.SetTimeMin(0.0, 100.0)
    P1.SetTimeMax(0.0, 800.0)
    P1.SetStateMin(0.0, 0.0, 0.0)
    P1.SetStateMax(21031.2, 21031.2, 21031.2)
    P1.SetStateMin(5.0, 5.0, 5.0)
    P1.SetStateMax(609.6, 609.6, 609.6)
    P1.SetStateMin(-0.7, -0.7, -0.7)
    P1.SetStateMax(0.7, 0.7, 0.7)
    P1.SetStateMin(10.0, 10.0, 10.0)
    P1.SetStateMax(20410.0, 20410.0, 20410.0)
    P1.SetcontrolMin(-math.pi / 4)
    P1.SetcontrolMax(math.pi / 4)
    for v in (h0, v0, gam0, m0, hf, vf, gamf):
        P1.SeteventMin(v)
    for v in (h0, v0, gam0, m0, hf, vf, gamf):
        P1.SeteventMax(v)
    P1.SetTimeGuess(0.0)
    P1.SetTimeGuess(400.0)
    for i, (a, b) in enumerate(((h0, hf), (v0, vf), (gam0, gamf), (m0, 0.9 * m0))):
        P1.SetStateGuess(i + 1, a)
        P1.SetStateGuess(i + 1, b)
    P1.SetControlGuess(1, 0.05)
    P1.SetControlGuess(1, 0.02)
    set_uniform_mesh(P1, n_intervals, nodes)
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_MIN_TIME_CLIMB,
                                             [Re, mu, S, g0, Isp, rho0, Hs, a0, a1, Tmax]))
    op.AddPhase(P1)
    return op


def quadrotor(n_intervals=8, nodes=8, pref=(1.0, -0.5, 1.5)):
    """BASELINE config 5 (one instance of the MPC sweep): nx=12, nu=4, no path/event constraints."""
    mass, g, arm, Ixx, Iyy, Izz, ktau = 1.2, 9.80665, 0.22, 0.011, 0.012, 0.021, 0.016
    wp, wv, wa, ww, wu = 4.0, 0.5, 1.0, 0.1, 0.05
    hov = mass * g / 4.0
    P1 = Phase(1, 12, 4, 0, 0, 0)
    P1.SetTimeMin(0.0, 2.0)
    P1.SetTimeMax(0.0, 2.0)
    init = [0.0] * 12
    for j in range(12):
        lim = 20.0 if j < 6 else (1.2 if j < 9 else 8.0)
        P1.SetStateMin(init[j], -lim, -lim)
        P1.SetStateMax(init[j], lim, lim)
    for _ in range(4):
        P1.SetcontrolMin(0.0)
        P1.SetcontrolMax(4.0 * hov)
    P1.SetTimeGuess(0.0)
    P1.SetTimeGuess(2.0)
    for j in range(12):
        P1.SetStateGuess(j + 1, init[j])
        P1.SetStateGuess(j + 1, pref[j] if j < 3 else 0.0)
    for j in range(4):
        P1.SetControlGuess(j + 1, hov)
        P1.SetControlGuess(j + 1, hov)
    set_uniform_mesh(P1, n_intervals, nodes)
    consts = [mass, g, arm, Ixx, Iyy, Izz, ktau, pref[0], pref[1], pref[2], wp, wv, wa, ww, wu]
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_QUADROTOR, consts))
    op.AddPhase(P1)
    return op


# --------------------------------------------------------------------------- static parameters (nq > 0), authored here
def param_sled(n_intervals=4, nodes=8, c0=0.5):
    """Minimum-time sled with a design parameter: x' = v, v' = p u, |u| <= 1, rest to rest over unit distance, cost
    tf + c0 p^2.  Bang-bang: tf = 2 / sqrt(p), so with c0 = 0.5 the optimum is p = 1, cost 2.5.  nx=2, nu=1, nq=1, ne=4."""
    P1 = Phase(1, 2, 1, 1, 0, 4)
    P1.SetTimeMin(0.0, 0.1)
    P1.SetTimeMax(0.0, 20.0)
    for _ in range(2):
        P1.SetStateMin(-5, -5, -5)
        P1.SetStateMax(5, 5, 5)
    P1.SetcontrolMin(-1.0)
    P1.SetcontrolMax(1.0)
    P1.SetparameterlMin(0.05)
    P1.SetparameterMax(10.0)
    for v in (0, 0, 1, 0):
        P1.SeteventMin(v)
        P1.SeteventMax(v)
    P1.SetTimeGuess(0.0)
    P1.SetTimeGuess(3.0)
    P1.SetStateGuess(1, 0.0)
    P1.SetStateGuess(1, 1.0)
    P1.SetStateGuess(2, 0.3)
    P1.SetStateGuess(2, 0.2)
    P1.SetControlGuess(1, 0.5)
    P1.SetControlGuess(1, -0.5)
    P1.SetparameterGuess(2.0)
    set_uniform_mesh(P1, n_intervals, nodes)
    op = OptimalProblem(1, 0, ProblemFunctor(RPM_PROBLEM_PARAM_SLED, [c0]))
    op.AddPhase(P1)
    return op


def param_oscillator(n_intervals=(5, 3), nodes=(6, 9)):
    """Damped oscillator with a stiffness parameter p0 and a weighting parameter p1 over two linked phases (nq = 2 each):
    dynamics, path constraint, running and terminal cost, events and the linkage (states AND parameters continuous across
    the phases) all depend on the parameters.  nx=2, nu=1, nq=2, nc=1, ne=2 / 1, 4 linkage constraints."""
    phases = []
    for ip in range(2):
        P = Phase(ip + 1, 2, 1, 2, 1, 2 if ip == 0 else 1)
        P.SetTimeMin(2.0 * ip, 2.0 * ip + 1.0)
        P.SetTimeMax(2.0 * ip, 2.0 * ip + 3.0)
        for _ in range(2):
            P.SetStateMin(-4, -4, -4)
            P.SetStateMax(4, 4, 4)
        P.SetcontrolMin(-3.0)
        P.SetcontrolMax(3.0)
        P.SetparameterlMin(0.2)
        P.SetparameterMax(5.0)
        P.SetparameterlMin(0.1)
        P.SetparameterMax(2.0)
        P.SetpathMin(-6.0)
        P.SetpathMax(6.0)
        if ip == 0:
            for v in (1.0, 0.0):
                P.SeteventMin(v)
                P.SeteventMax(v)
        else:
            P.SeteventMin(0.0)
            P.SeteventMax(1.5)
        P.SetTimeGuess(2.0 * ip)
        P.SetTimeGuess(2.0 * ip + 2.0)
        P.SetStateGuess(1, 1.0 - 0.4 * ip)
        P.SetStateGuess(1, 0.6 - 0.4 * ip)
        P.SetStateGuess(2, -0.3)
        P.SetStateGuess(2, -0.1)
        P.SetControlGuess(1, 0.2)
        P.SetControlGuess(1, -0.1)
        P.SetparameterGuess(1.3)
        P.SetparameterGuess(0.7)
        set_uniform_mesh(P, n_intervals[ip], nodes[ip])
        phases.append(P)
    lk = Linkage(1, 1, 2)
    for _ in range(4):
        lk.SetLinkMin(0)
        lk.SetLinkMax(0)
    op = OptimalProblem(2, 1, ProblemFunctor(RPM_PROBLEM_PARAM_OSC, [0.4, 0.05]))
    for P in phases:
        op.AddPhase(P)
    op.AddLinkage(lk)
    return op


# --------------------------------------------------------------------------- configs + iterates
def config(name):
    """BASELINE.json configs by short name -> OptimalProblem."""
    if name == "brachistochrone":
        return brachistochrone(1, 10)
    if name == "climb":
        return min_time_climb(16, 16)
    if name == "launch":
        return launch(64, 16)
    if name == "launch_default":
        return launch()
    if name == "hypersensitive":
        mesh, nodes = hp_mesh(4096)
        return hypersensitive(mesh, nodes)
    if name == "hypersensitive_uniform":
        return hypersensitive([-1.0 + 2.0 * i / 256 if i < 256 else 1.0 for i in range(257)], [16] * 256)
    if name == "quadrotor":
        return quadrotor(8, 8)
    if name == "bryson_denham":
        return bryson_denham()
    if name == "param_sled":
        return param_sled()
    if name == "param_oscillator":
        return param_oscillator()
    raise KeyError(name)


def seeded_iterate(x_guess, x_l, x_u, seed, mode="perturb", rel=1e-3):
    """Synthetic NLP iterates: "perturb" = guess * (1 + rel*U(-1,1)) with exact zeros
    nudged off zero, "uniform" = uniform inside finite bounds (guess where a bound is infinite)."""
    rng = np.random.RandomState(seed)
    x_guess = np.asarray(x_guess, dtype=np.float64)
    if mode == "perturb":
        u = rng.uniform(-1.0, 1.0, size=x_guess.shape)
        x = x_guess * (1.0 + rel * u)
        z = x_guess == 0.0
        x[z] = rel * u[z]
        return x
    lo = np.where(np.isfinite(x_l), x_l, x_guess - 1.0)
    hi = np.where(np.isfinite(x_u), x_u, x_guess + 1.0)
    return lo + (hi - lo) * rng.uniform(0.0, 1.0, size=x_guess.shape)
