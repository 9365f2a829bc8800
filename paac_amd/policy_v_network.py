"""Actor-critic heads + loss (mirrors reference policy_v_network.py:4-64).

Keeps the attribute names PAACLearner feeds and fetches (paac.py:20-23,140-142,157-160; actor_learner.py:44):
`output_layer_pi`, `output_layer_v`, `critic_target_ph`, `adv_actor_ph`, `selected_action_ph`, `loss`.
The maths (softmax head, linear critic, log(pi+1e-30), entropy, actor/critic means, loss = 5*(actor+critic))
runs in csrc/heads.h: heads_fwd_kernel / heads_bwd_kernel.
"""
import numpy as np

from .networks import NatureNetwork, Network, NIPSNetwork, Placeholder


class PolicyVNetwork(Network):

    def __init__(self, conf):
        super(PolicyVNetwork, self).__init__(conf)
        self.entropy_regularisation_strength = conf['entropy_regularisation_strength']
        self.critic_target_ph = Placeholder('target', np.float32, [None])
        self.adv_actor_ph = Placeholder('advantage', np.float32, [None])
        self.output_layer_pi = Placeholder('actor_output_policy')
        self.output_layer_v = Placeholder('critic_output')
        self.log_output_layer_pi = Placeholder('actor_output_log_policy')
        self.output_layer_entropy = Placeholder('entropy')
        self.loss = Placeholder('loss')
        self._allocate()


class NIPSPolicyVNetwork(PolicyVNetwork, NIPSNetwork):
    pass


class NaturePolicyVNetwork(PolicyVNetwork, NatureNetwork):
    pass
