"""Simulation constants: same keys and defaults as the reference's METADATA dict
(Simulation/constants.py:26-57), so learners that read sim.METADATA keep working.
ForestFire() without arguments works on THIS dict, like the reference (one module-global dict that step()
mutates, forest_fire.py:40-43); BatchedForestFire and ForestFire(metadata=..., size=...) take a private copy."""

SIZE = 10
A_SPEED = 1

METADATA = {
    # reward measure (constants.py:32-34)
    "death_penalty": -1000 * A_SPEED,
    "contained_bonus": 1000 * A_SPEED,
    "default_reward": -1,
    # simulation constants (constants.py:37-46)
    "width": SIZE,
    "height": SIZE,
    "wind": [0.54, (0, 0)],
    "debug": 1,
    "n_actions": 4,
    "a_speed": A_SPEED,
    "a_speed_iter": A_SPEED,
    "make_rivers": False,
    "containment_wins": False,
    "allow_dig_toggle": False,
    # DQN parameters (constants.py:49-56) - carried for the learners, unused by the step
    "memory_size": 20000,
    "max_eps": 1.0,
    "min_eps": 0.01,
    "eps_decay_rate": 0.005,
    "gamma": 0.999,
    "alpha": 0.005,
    "target_update": 20,
    "batch_size": 32,
}
