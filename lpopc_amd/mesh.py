"""Mesh refinement driver: the host-side mirror of lpopc's MeshRefiner.

Reference: Core/LpMeshRefiner.h:36-92 and Core/LpMeshRefiner.cpp:62-90 (grid counter, mesh history, method choice
from the option list), Core/LpPhMeshRefineAlg.cpp (the ph method), Core/LpLiuHpMeshRefineAlg.cpp (hp-Liu),
Core/Nlp2OPConverter.cpp:149-193 (the extracted
solution becomes the next mesh's guess).  The per-node work runs on the device behind rpm_solution_error /
rpm_ph_refine_mesh / rpm_hpliu_refine; this class only keeps the reference's bookkeeping and error behaviour.
"""
import numpy as np
from .problem import LpopcException


class LpMesh:
    """Core/LpMeshRefineImpletation.hpp: the mesh of one phase."""

    def __init__(self, meshpoints, nodesPerInterval):
        self.meshpoints = list(meshpoints)
        self.nodesPerInterval = list(nodesPerInterval)


class MeshRefiner:
    def __init__(self, options):
        self.tol_ = options.GetNumericValue("desired-relative-error")
        self.method_ = options.GetStringValue("mesh-refine-methods")
        self.gridmaxnum_ = options.GetIntegerValue("max-grid-num")
        self.Nmax_ = options.GetIntegerValue("Nmax")
        self.Nmin_ = options.GetIntegerValue("Nmin")
        self.R_ = options.GetNumericValue("R")
        self.grid_ = 0
        self.meshhistory = []
        self._hpliu = None   # LiuHpMeshRefineAlg keeps its own histories across meshes (Core/LpMeshRefiner.h:54-61)

    def CurrentGrid(self):
        return self.grid_

    def RefineMesh(self, engine, optpro, x=None):
        """One refinement pass on the solution x (default: the one stored by finalize_solution).  Installs the new
        mesh in `optpro` and returns NoMoreRefine; the caller builds a new engine for the new mesh."""
        if self.grid_ > self.gridmaxnum_:
            raise LpopcException("The problem reach the max number of refine grid,but hasn't reach the derized error tolrance")
        nph = optpro.GetPhaseNum()
        if self.grid_ == 0:
            self.meshhistory.append([LpMesh(optpro.GetPhase(i).GetMeshPoints(), optpro.GetPhase(i).GetNodesPerInterval())
                                     for i in range(nph)])
        no_more, newmesh = True, []
        if self.method_ == "hp-Liu":
            if self._hpliu is None:
                from .engine import HpLiuRefiner
                self._hpliu = HpLiuRefiner(nph, self.tol_, self.Nmax_, self.R_)
            no_more, meshes = self._hpliu.refine(engine, x=x)
            newmesh = [LpMesh(mesh.tolist(), [int(v) for v in nodes]) for mesh, nodes in meshes]
        else:
            for i in range(nph):
                done, mesh, nodes, _ = engine.ph_refine_mesh(i, self.tol_, self.Nmin_, self.Nmax_, x=x)
                no_more = no_more and done
                newmesh.append(LpMesh(mesh.tolist(), [int(v) for v in nodes]))
        for i, mesh in enumerate(newmesh):   # the reference rewrites the mesh even when nothing changed
            ph = optpro.GetPhase(i)
            ph.meshpoints = list(mesh.meshpoints)
            ph.nodesperinterval = list(mesh.nodesPerInterval)
        if not no_more:
            self.grid_ += 1
            self.meshhistory.append(newmesh)
        return no_more


def install_guess(engine, optpro, x=None, lam=None):
    """Nlp2OpControl's tail (Core/Nlp2OPConverter.cpp:160-193): the extracted time / state / control arrays of every
    phase become that phase's guess for the next mesh."""
    xs = np.asarray(x if x is not None else engine.get_solution()[0], dtype=np.float64)
    off = 0
    for i in range(optpro.GetPhaseNum()):
        ph = optpro.GetPhase(i)
        r = engine.nlp2op_control(i, x=x, lam=lam)
        M = r["time"].size
        ph.vtimeguess = [float(v) for v in r["time"]]
        nx, nu = r["state"].size // M, r["control"].size // M
        ph.vstateguess = [[float(v) for v in r["state"][s * M:(s + 1) * M]] for s in range(nx)]
        ph.vcontrolguess = [[float(v) for v in r["control"][j * M:(j + 1) * M]] for j in range(nu)]
        nq = ph.get_optimal_info()[2]
        p0 = off + nx * M + nu * (M - 1) + 2                      # [X | U | t0 tf | p], Core/LpBoundsChecker.cpp:51-138
        ph.vparameterguess = [float(v) for v in xs[p0:p0 + nq]]   # vparameterguess, Nlp2OPConverter.cpp:185-193
        off = p0 + nq
