"""Host-side mirror of lpopc's problem-setup API (same class and method names, same argument
meaning, same error behaviour), reference Core/LpOptimalProblem.hpp:30-326.

The reference's `FunctionWrapper` (Core/LpFunctionWrapper.h:50-69) is a host C++ class with
vectorised Armadillo callbacks and cannot run on a GPU; its place is taken by a
`ProblemFunctor`: the id of a pointwise device functor compiled into the native library plus
the problem constants the reference keeps in globals (e.g. CONSTANTS, example/launch/Launch.cpp:47-74).
"""


class LpopcException(Exception):
    """Common/LpException.hpp:14 — raised where the reference does LP_THROW_EXCEPTION."""


class Limit:
    """Core/LpOptimalProblem.hpp:18-29"""

    def __init__(self, state0, state, statef):
        self.state = [float(state0), float(state), float(statef)]

    def GetLimit(self):
        return tuple(self.state)


class ProblemFunctor:
    """Stands where shared_ptr<FunctionWrapper> stands in the reference."""

    def __init__(self, problem_id, consts=(), library=None):
        self.problem_id = int(problem_id)
        self.consts = [float(c) for c in consts]
        # path of the native library that holds the functor: None = the package's librpm_hip.so (built-in problems);
        # for RPM_PROBLEM_USER the library lpopc_amd.userproblem.build() compiled from the user's functor header
        self.library = library


class Phase:
    """Core/LpOptimalProblem.hpp:30-240"""

    def __init__(self, phase_index, statenum, controlnum, parameternum, pathnum, eventnum):
        self.phase_index_ = phase_index
        self.statenum_ = statenum
        self.controlnum_ = controlnum
        self.parameternum_ = parameternum
        self.pathnum_ = pathnum
        self.eventnum_ = eventnum
        self.hasduration_ = False
        self.vstatemin, self.vstatemax = [], []
        self.vcontrolmin, self.vcontrolmax = [], []
        self.vparametermin, self.vparametermax = [], []
        self.vpathmin, self.vpathmax = [], []
        self.veventmin, self.veventmax = [], []
        self.vtimemin = self.vtimemax = self.vduration = None
        self.vtimeguess, self.vstateguess, self.vcontrolguess, self.vparameterguess = [], [], [], []
        self.meshpoints, self.nodesperinterval = [], []
        self.nodes = 0

    def get_optimal_info(self):
        return (self.statenum_, self.controlnum_, self.parameternum_, self.pathnum_, self.eventnum_)

    def SetTimeMin(self, t0, tf):
        self.vtimemin = Limit(t0, 0, tf)

    def GetTimeMin(self):
        return self.vtimemin.state[0], self.vtimemin.state[2]

    def SetTimeMax(self, t0, tf):
        self.vtimemax = Limit(t0, 0, tf)

    def GetTimeMax(self):
        return self.vtimemax.state[0], self.vtimemax.state[2]

    def SetStateMin(self, state0, state, statef):
        self.vstatemin.append(Limit(state0, state, statef))

    def GetstateMin(self):
        return self.vstatemin

    def SetStateMax(self, state0, state, statef):
        self.vstatemax.append(Limit(state0, state, statef))

    def GetstateMax(self):
        return self.vstatemax

    def SetcontrolMin(self, v):
        self.vcontrolmin.append(float(v))

    def GetcontrolMin(self):
        return self.vcontrolmin

    def SetcontrolMax(self, v):
        self.vcontrolmax.append(float(v))

    def GetcontrolMax(self):
        return self.vcontrolmax

    def SetparameterlMin(self, v):  # (sic) Core/LpOptimalProblem.hpp:97
        self.vparametermin.append(float(v))

    def GetparameterMin(self):
        return self.vparametermin

    def SetparameterMax(self, v):
        self.vparametermax.append(float(v))

    def GetparameterMax(self):
        return self.vparametermax

    def SetpathMin(self, v):
        self.vpathmin.append(float(v))

    def GetpathMin(self):
        return self.vpathmin

    def SetpathMax(self, v):
        self.vpathmax.append(float(v))

    def GetpathMax(self):
        return self.vpathmax

    def SeteventMin(self, v):
        self.veventmin.append(float(v))

    def GeteventMin(self):
        return self.veventmin

    def SeteventMax(self, v):
        self.veventmax.append(float(v))

    def GeteventMax(self):
        return self.veventmax

    def SetDuration(self, durationmin, durationmax):
        self.vduration = Limit(durationmin, 0, durationmax)
        self.hasduration_ = True

    def Getduration(self):
        return self.vduration.state[0], self.vduration.state[2]

    def HasDuration(self):
        return self.hasduration_

    def SetTimeGuess(self, guess):
        self.vtimeguess.append(float(guess))

    def GetTimeGuess(self):
        return self.vtimeguess

    def SetStateGuess(self, stateindex, guess):  # 1-based, :135-143
        if len(self.vstateguess) >= stateindex:
            self.vstateguess[stateindex - 1].append(float(guess))
        elif stateindex == len(self.vstateguess) + 1:
            self.vstateguess.append([float(guess)])

    def GetStateGuess(self):
        return self.vstateguess

    def SetControlGuess(self, controlindex, guess):
        if len(self.vcontrolguess) >= controlindex:
            self.vcontrolguess[controlindex - 1].append(float(guess))
        elif controlindex == len(self.vcontrolguess) + 1:
            self.vcontrolguess.append([float(guess)])

    def GetControlGuess(self):
        return self.vcontrolguess

    def SetparameterGuess(self, guess):
        self.vparameterguess.append(float(guess))

    def GetparameterGuess(self):
        return self.vparameterguess

    def SetMeshPoints(self, meshpoint):
        self.meshpoints.append(float(meshpoint))

    def GetMeshPoints(self):
        return self.meshpoints

    def SetNodesPerInterval(self, nodes):
        self.nodesperinterval.append(int(nodes))

    def GetNodesPerInterval(self):
        return self.nodesperinterval

    def SetTotalNodes(self, n):
        self.nodes = n

    def GetTotalNodes(self):
        return self.nodes


class Linkage:
    """Core/LpOptimalProblem.hpp:242-279"""

    def __init__(self, ipair, left, right):
        self.pairindex, self.leftphase, self.rightphase = ipair, left, right
        self.linkmin, self.linkmax = [], []

    def SetLinkMin(self, v):
        self.linkmin.append(float(v))

    def SetLinkMax(self, v):
        self.linkmax.append(float(v))

    def GetLinkageMin(self):
        return self.linkmin

    def GetLinkageMax(self):
        return self.linkmax

    def LeftPhase(self):
        return self.leftphase - 1

    def RightPhase(self):
        return self.rightphase - 1


class OptimalProblem:
    """Core/LpOptimalProblem.hpp:281-326"""

    def __init__(self, numphase, numlinkage, userfun):
        self.numphase_, self.numlink_, self.userfunction_ = numphase, numlinkage, userfun
        self.Phases_, self.Linkage_ = [], []

    def AddPhase(self, phase):
        self.Phases_.append(phase)

    def AddLinkage(self, linkage):
        self.Linkage_.append(linkage)

    def GetPhase(self, phaseindex):
        if not (0 <= phaseindex < self.numphase_) or phaseindex >= len(self.Phases_):
            raise LpopcException("The phase index is out of rang in Function 'GetPhase' ")
        return self.Phases_[phaseindex]

    def GetLinkage(self, linkindex):
        if not (0 <= linkindex < len(self.Linkage_)):
            raise LpopcException("The linkage index is out of rang in Function 'GetLinkage' ")
        return self.Linkage_[linkindex]

    def GetPhaseNum(self):
        return self.numphase_

    def GetLinkageNum(self):
        return self.numlink_

    def GetOpimalProblemFuns(self):  # (sic) :314
        return self.userfunction_


class Options:
    """The 13 registered options of the reference with their defaults (Core/LpOptDerive.hpp:29-36,
    Core/LpMeshRefiner.h:67-80, Core/LpNLPWrapper.hpp:69-76); setters as Common/LpOptionList.hpp:175-220."""

    _DEFAULTS = {
        "finite-difference-tol": 1e-6,
        "first-derive": "finite-difference",
        "analytic-derive-check": "no",
        "analytic-derive-check-tol": 1e-7,
        "mesh-refine-methods": "ph",
        "max-grid-num": 10,
        "desired-relative-error": 1e-6,
        "Nmax": 16,
        "Nmin": 4,
        "R": 1.2,
        "hessian-approximation": "limited-memory",
        "Ipopt-tol": 1e-6,
        "auto-scale": "no",
    }
    _CHOICES = {
        "first-derive": ("finite-difference", "analytic"),
        "analytic-derive-check": ("yes", "no"),
        "mesh-refine-methods": ("ph", "hp-Liu"),
        "hessian-approximation": ("limited-memory", "exact"),
        "auto-scale": ("yes", "no"),
    }

    def __init__(self):
        self._v = dict(self._DEFAULTS)

    def _check(self, name, typ):
        if name not in self._v:
            raise LpopcException("Tried to set Option: %s. It is not a valid option." % name)
        if not isinstance(self._DEFAULTS[name], typ):
            raise LpopcException("Tried to set Option: %s with the wrong type." % name)

    def SetStringValue(self, name, value):
        self._check(name, str)
        if value not in self._CHOICES[name]:
            raise LpopcException("Setting: \"%s\" is not a valid setting for Option: %s." % (value, name))
        self._v[name] = value
        return True

    def SetNumericValue(self, name, value):
        self._check(name, float)
        self._v[name] = float(value)
        return True

    def SetIntegerValue(self, name, value):
        self._check(name, int)
        self._v[name] = int(value)
        return True

    def GetStringValue(self, name):
        return self._v[name]

    def GetNumericValue(self, name):
        return self._v[name]

    def GetIntegerValue(self, name):
        return self._v[name]


def apply_mesh_defaults(phase):
    """MeshRefiner::SetAndCheckMesh (Core/LpMeshRefiner.cpp:10-62): default mesh [-1,1],
    default 20 nodes per interval (the log line says 10, the code pushes 20 — SURVEY B-16)."""
    mesh = list(phase.GetMeshPoints())
    nodes = list(phase.GetNodesPerInterval())
    if len(mesh) == 0:
        if nodes:
            k = len(nodes)
            mesh = [-1.0 + 2.0 * i / k for i in range(k + 1)]
            mesh[-1] = 1.0
        else:
            mesh = [-1.0, 1.0]
    elif len(mesh) == 1:
        raise LpopcException("MeshRefinement need at least two  meshPoints,but there's only one in phase%d"
                             % phase.phase_index_)
    elif mesh[0] != -1 or mesh[-1] != 1:
        raise LpopcException("meshPoints must span -1 to +1 in phase%d" % phase.phase_index_)
    if len(nodes) == 0:
        nodes = [20] * (len(mesh) - 1)
    elif len(mesh) != len(nodes) + 1:
        raise LpopcException("Number of nodesPerInterval must match number of mesh intervals in phase%d"
                             % phase.phase_index_)
    return mesh, nodes
