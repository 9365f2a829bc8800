"""paac_amd -- MI355X-native drop-in for the rollout/update hot path of arjunchandra/paac.

Same plugin surface as the reference (BaseEnvironment / EnvironmentCreator / network_creator ->
PolicyVNetwork / PAACLearner(network_creator, environment_creator, args).train()), with the TensorFlow-1
graph, the multiprocessing shared-memory batching and the host-side numpy maths replaced by hand-written
HIP kernels for gfx950 behind a C-ABI (include/paac_hip.h, paac_amd/libpaac_hip.so).
There is no CPU path: importing the ops without the built library raises.
"""
__version__ = "0.1.0"
