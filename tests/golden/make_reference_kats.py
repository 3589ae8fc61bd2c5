#!/usr/bin/env python3
"""Transcribes the known-answer data of the reference's own tests into JSON fixtures.

The reference (swarth100/dynfu) has no Python and cannot be built here, so nothing is
imported or executed from it: the numbers below are the literals of
  test/quaternion_test.cpp            (21 DualQuaternionTest cases, tolerance 1e-4, :40)
  test/opt_optimisation_test.cpp      (8 OptTest scenes, tolerance 1e-3, :94)
written down as data (inputs + expected outputs), each with the file:line it comes from.
Run:  python tests/golden/make_reference_kats.py   -> dq_kat.json, opt_scenes.json
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# The reference computes its angle constants in float: float RAD30 = M_PI / 6, ... (:30-38).
# RAD120/RAD150 are unused by any assertion.
import numpy as np  # noqa: E402

f32 = lambda x: float(np.float32(x))
RAD = {"RAD180": f32(math.pi), "RAD90": f32(math.pi / 2), "RAD60": f32(math.pi / 3), "RAD45": f32(math.pi / 4),
       "RAD30": f32(math.pi / 6), "RAD15": f32(math.pi / 12), "0": 0.0}

# fixture objects (quaternion_test.cpp:42-48): Euler ctor args (yaw, pitch, roll, x, y, z)
FIX = {
    "dq90": ["RAD90", "RAD90", "RAD90", 0, 0, 0],
    "dq60": ["RAD60", "RAD60", "RAD60", 0, 0, 0],
    "dq45": ["RAD45", "RAD45", "RAD45", 0, 0, 0],
    "dq30Rot": ["RAD30", "RAD30", "RAD30", 0, 0, 0],
    "dq0": ["0", "0", "0", 0, 0, 0],
    "dq30": ["0", "RAD30", "0", 0, 0, 100.0],
    # locals used by several tests
    "dqA": ["RAD30", "RAD45", "RAD30", 30, 20, 10],   # :143, :186, :245, :290
    "dq30New": ["0", "RAD30", "0", 0, 0, 0],          # :94, :378, ...
    "dqT": ["0", "0", "0", 1.0, 0, 0],                # :349
    "dq90T": ["RAD90", "RAD90", "RAD90", 1.0, 0, 0],  # :363
}

# Each case: expr is a tiny prefix program evaluated by tests/test_oracle_dq.py:
#   ["fix", name] | ["add",a,b] | ["sub",a,b] | ["mul",a,b] | ["scale",a,s] | ["normalize",a] | ["rodrigues",[x,y,z]]
# expect_real / expect_dual: 4 numbers or null (component not asserted by the reference).
CASES = [
    dict(name="TestReal", line="57-65", expr=["fix", "dq45"],
         expect_real=[0.8446231020115715, 0.19134170284356308, 0.4619399539487806, 0.19134170284356303]),
    dict(name="TestDual", line="70-91", expr=["fix", "dq30"], expect_real=[0.9659, 0.0, 0.2588, 0.0],
         expect_dual=[0.0, -12.9409, 0.0, 48.2962]),
    dict(name="TestFromRodrigues/30", line="93-120", expr=["rodrigues", [0.0, 0.267949192431123, 0.0]],
         expect_real_of=["fix", "dq30New"]),
    dict(name="TestFromRodrigues/45", line="93-120",
         expr=["rodrigues", [0.226540919660986, 0.546918160678027, 0.226540919660986]],
         expect_real_of=["fix", "dq45"]),
    dict(name="TestFromRodrigues/90", line="93-120", expr=["rodrigues", [0.0, 1.0, 0.0]],
         expect_real_of=["fix", "dq90"]),
    dict(name="TestSum", line="123-140", expr=["add", ["fix", "dq45"], ["fix", "dq30"]],
         expect_real=[1.8105, 0.1913, 0.7208, 0.1913], expect_dual=[0.0, -12.9410, 0.0, 48.2963]),
    dict(name="TestSumAssign", line="160-180", expr=["add", ["fix", "dqA"], ["fix", "dq30"]],
         expect_real=[1.8536, 0.1353, 0.6778, 0.1353], expect_dual=[-6.8953, -0.3683, 7.5233, 57.6655]),
    dict(name="TestDiff", line="183-201", expr=["sub", ["fix", "dq45"], ["fix", "dq30"]],
         expect_real=[-0.1213, 0.1913, 0.2031, 0.1913], expect_dual=[0.0, 12.9410, 0.0, -48.2963]),
    dict(name="TestDiffAssign", line="203-223", expr=["sub", ["fix", "dqA"], ["fix", "dq30"]],
         expect_real=[-0.0783, 0.1353, 0.1601, 0.1353], expect_dual=[-6.8953, 25.5137, 7.5233, -38.9271]),
    dict(name="TestScale", line="226-243", expr=["scale", ["fix", "dq30"], 0.30],
         expect_real_of=["fix", "dq30"], expect_dual=[0.0, -3.8823, 0.0, 14.4889]),
    dict(name="TestScaleAssign", line="245-265", expr=["scale", ["fix", "dqA"], 0.30],
         expect_real_of=["fix", "dqA"], expect_dual=[-2.0686, 3.7718, 2.2570, 2.8108]),
    dict(name="TestMul", line="268-286", expr=["mul", ["fix", "dq30"], ["fix", "dq45"]],
         expect_real=[0.6963, 0.2343, 0.6648, 0.1353], expect_dual=[-6.7650, -33.2402, 11.7172, 34.8142]),
    dict(name="TestMulAssign", line="289-308", expr=["mul", ["fix", "dqA"], ["fix", "dq30"]],
         expect_real=[0.7490, 0.0957, 0.6344, 0.1657], expect_dual=[-13.3911, 18.4657, -2.8031, 60.5945]),
    dict(name="TestNormalize", line="311-330", expr=["normalize", ["add", ["fix", "dq45"], ["fix", "dq30"]]],
         expect_real=[0.9203, 0.0973, 0.3663, 0.0973], expect_dual=[0.0, -12.9410, 0.0, 48.2963]),
]

# transformVertex cases: dq expr, vertex, expected
TRANSFORMS = [
    dict(name="TestDoNotTransform", line="333-340", expr=["fix", "dq0"], v=[0, 0, 1], expect=[0, 0, 1]),
    dict(name="TestRotate", line="343-350", expr=["fix", "dq90"], v=[0, 0, 1], expect=[1, 0, 0]),
    dict(name="TestTranslate", line="353-362", expr=["fix", "dqT"], v=[0, 0, 1], expect=[1, 0, 1]),
    dict(name="TestTranslateAndRotate", line="365-374", expr=["fix", "dq90T"], v=[0, 0, 1], expect=[2, 0, 0]),
]

# TestComposeRotations (:143-157): T(dq90*dq90)(v) == T(dq90)(T(dq90)(v))
COMPOSE = dict(name="TestComposeRotations", line="143-157", a=["fix", "dq90"], b=["fix", "dq90"], v=[0, 0, 1])

# angle getters (:377-435): expected in terms of the RAD table
ANGLES = [
    dict(name="RollTest", line="377-387", getter="roll", cases=[["dq30New", "0"], ["dq45", "RAD45"], ["dq90", "RAD90"]]),
    dict(name="PitchTest", line="390-398", getter="pitch", cases=[["dq30", "RAD30"], ["dq45", "RAD45"], ["dq90", "RAD90"]]),
    dict(name="YawTest", line="401-411", getter="yaw", cases=[["dq30New", "0"], ["dq45", "RAD45"], ["dq90", "RAD90"]]),
    # ConvertToEulerAnglesTest (:414-435) asserts the same getters through getEulerAngles()
]
RODRIGUES_GET = dict(name="ConvertToRodriguesTest", line="438-456", cases=[
    ["dq30New", [0.0, 0.267949192431123, 0.0]],
    ["dq45", [0.226540919660986, 0.546918160678027, 0.226540919660986]],
    ["dq90", [0.0, 1.0, 0.0]],
])
TOSTRING = dict(name="TestToString", line="458-462", fix="dq30",
                expect="real: (0.965926,0,0.258819,0)\ndual: (0,-12.941,0,48.2963)\n")

json.dump(dict(source="swarth100/dynfu test/quaternion_test.cpp", tol=1e-4, rad=RAD, fixtures=FIX, cases=CASES,
               transforms=TRANSFORMS, compose=COMPOSE, angles=ANGLES, rodrigues_get=RODRIGUES_GET, tostring=TOSTRING),
          open(os.path.join(HERE, "dq_kat.json"), "w"), indent=1)

# ----------------------------------------------------------------------------------------
# OptTest scenes (test/opt_optimisation_test.cpp)
G1 = [[3, 1, -1], [1, 1, 1], [-1, 2, 3], [-1, -1, 1], [-2, -1, -1], [2, -1, -3], [-1, 1, -1], [2, 1, 1]]  # :56-63
G2 = [[10, 10, 10], [9, 11.1, 10], [10, 9, 10], [10, 12, 9], [9, 11, 10], [12, 10, 9], [9, 9, 12], [10.5, 9, 9],
      [10.5, 12, 12], [11, 11, 10.9]]  # :66-75
diag = lambda *xs: [[x, x, x] for x in xs]
S5 = diag(-3, -2, 0.01, 2, 3)            # :283-287
T5 = diag(-2.99, -1.99, 0.02, 2.01, 3.01)  # :297-301
S5b = diag(-3, -2, 0.04, 2, 3)           # :457-461
T5b1 = diag(-2.99, -1.99, 0.05, 2.01, 3.01)  # :471-475
T5b2 = diag(-2.98, -1.98, 0.06, 2.02, 3.02)  # :503-507
T5b3 = diag(-2.96, -1.96, 0.09, 2.04, 3.05)  # :603-607
S10 = S5 + diag(12, 11, 10, 10.5, 11.5)  # :382-401
T10 = T5 + diag(11.99, 10.99, 9.99, 10.51, 11.49)  # :411-430

# ops:  ["solve", canon, live]            CombinedSolver(...).initializeProblemInstance(canon, live); solveAll()
#       ["assert_warp", src, expected]    for v in src: calcDQB(v).transformVertex(v) ~= expected   (|d| <= tol per axis)
#       ["warp", src, dst]                dst = warpfield.warpToLive(src)
SCENES = [
    dict(name="SingleVertexOneGroupOfDeformationNodesTest", line="212-240", nodes="g1",
         sets={"S": [[0, 0.04, 0]], "T": [[0.01, 0.03, 0]]},
         ops=[["solve", "S", "T"], ["assert_warp", "S", "T"]]),
    dict(name="TwoVerticesOneNotMovingOneGroupOfDeformationNodesTest", line="243-277", nodes="all",
         sets={"S": [[0, 0.05, 1], [2, 2, 2]], "T": [[0.01, 0.04, 1.01], [2, 2, 2]]},
         ops=[["solve", "S", "T"], ["assert_warp", "S", "T"]]),
    dict(name="MultipleVerticesOneGroupOfDeformationNodesTest", line="280-326", nodes="g1",
         sets={"S": S5, "T": T5}, ops=[["solve", "S", "T"], ["assert_warp", "S", "T"]]),
    dict(name="OneGroupOfVerticesTwoGroupsOfDeformationNodes", line="329-375", nodes="all",
         sets={"S": S5, "T": T5}, ops=[["solve", "S", "T"], ["assert_warp", "S", "T"]]),
    dict(name="TwoGroupsOfVerticesTwoGroupsOfDeformationNodes", line="378-451", nodes="all",
         sets={"S": S10, "T": T10}, ops=[["solve", "S", "T"], ["assert_warp", "S", "T"]]),
    dict(name="MultipleVerticesOneGroupOfDeformationNodesWarpTwiceTest", line="454-527", nodes="g1",
         sets={"S": S5b, "T1": T5b1, "T2": T5b2},
         ops=[["solve", "S", "T1"], ["assert_warp", "S", "T1"], ["warp", "S", "W1"], ["solve", "W1", "T2"],
              ["assert_warp", "S", "T2"]]),
    dict(name="MultipleVerticesOneGroupOfDeformationNodesWarpThriceTest", line="530-629", nodes="g1",
         sets={"S": S5b, "T1": T5b1, "T2": T5b2, "T3": T5b3},
         ops=[["solve", "S", "T1"], ["assert_warp", "S", "T1"], ["warp", "S", "W1"], ["solve", "W1", "T2"],
              ["assert_warp", "S", "T2"], ["warp", "W1", "W2"], ["solve", "W2", "T3"], ["assert_warp", "W1", "T3"]]),
    dict(name="MultipleVerticesOneGroupOfDeformationNodesWarpAndReverseTest", line="632-698", nodes="g1",
         sets={"S": S5b, "T": T5b1},
         ops=[["solve", "S", "T"], ["assert_warp", "S", "T"], ["solve", "T", "S"], ["assert_warp", "S", "S"]]),
]
json.dump(dict(source="swarth100/dynfu test/opt_optimisation_test.cpp", tol=1e-3, knn=8, dg_w=2.0,
               params=dict(numIter=32, nonLinearIter=16, linearIter=256, useOpt=False, useOptLM=True, earlyOut=True,
                           optDoublePrecision=True),  # :38-44
               tukeyOffset=4.652, psi_data=1e-2, lambda_=0.0, psi_reg=1e-4,  # :115-122
               nodes=dict(g1=G1, g2=G2), scenes=SCENES),
          open(os.path.join(HERE, "opt_scenes.json"), "w"), indent=1)
print("wrote dq_kat.json, opt_scenes.json")
