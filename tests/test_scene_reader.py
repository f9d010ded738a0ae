"""MJCF scene reader without MuJoCo: the data the reference adapter extracts, pinned by
the known answers of upstream tests/unit/simulation/test_mujoco_sim.py.  Host-side only (no GPU call)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

SCENE = os.path.join(GOLDEN, "lab_scene_min.xml")


@pytest.fixture(scope="module")
def sim():
    from uav_ac.simulation.mujoco_sim import MujocoSimulation
    return MujocoSimulation(SCENE)


def test_scene_matches_reference_known_answers(sim):
    g = load_golden("fixed_missions.npz")
    assert sim.start_position == pytest.approx([1.0, 7.0, -0.021])              # upstream :23-35
    assert sim.goal_position == pytest.approx([23.0, 7.0, -2.0])
    assert sim.space_limits == pytest.approx(np.array([[0.0, 0.0, -6.0], [24.0, 14.0, 0.0]]))
    assert sim.mission_waypoints == pytest.approx(g["lab_wp"])                    # upstream :38-58
    assert sim.obstacles[0] == pytest.approx([3.7, 4.3, 4.0, 10.0, -3.4, -2.8])   # upstream :239-248
    assert sim.obstacles == pytest.approx(g["lab_aabbs"])
    q = sim.quad
    assert (q.m, q.dt, q.g, q.l, q.kf, q.kappa) == pytest.approx((0.5, 0.001, 9.81, 0.120208, 1.0, 0.016))
    assert (q.i_x, q.i_y, q.i_z) == pytest.approx((0.0023, 0.0023, 0.0046))
    assert (q.min_thrust, q.max_thrust, q.max_tilt_angle) == pytest.approx((0.1, 4.5, 0.7))
    assert list(sim.rotor_spin_directions) == [1, -1, 1, -1]
    assert q.position == pytest.approx([1.0, 7.0, -0.021]) and q.X[3] == 1.0


def test_enu_to_ned_known_answer():
    from uav_ac.simulation.mujoco_sim import mujoco_to_ned_state
    s = mujoco_to_ned_state(np.array([1.0, -2.0, 3.0]), np.array([np.sqrt(0.5), 0, 0, np.sqrt(0.5)]),
                            np.array([4.0, -5.0, 6.0, 0.1, -0.2, 0.3]))                # upstream :61-74
    assert s[:3] == pytest.approx([1, 2, -3]) and s[3:7] == pytest.approx([np.sqrt(0.5), 0, 0, -np.sqrt(0.5)])
    assert s[7:10] == pytest.approx([4, 5, -6]) and s[10:13] == pytest.approx([0.1, 0.2, -0.3])
    with pytest.raises(ValueError):
        mujoco_to_ned_state(np.zeros(3), np.zeros(4), np.zeros(6))
    with pytest.raises(ValueError):
        mujoco_to_ned_state(np.array([np.nan, 0, 0]), np.array([1.0, 0, 0, 0]), np.zeros(6))


@pytest.mark.parametrize("old, new, msg", [
    ('<site name="goal" pos="23 -7 2"/>', "", "goal"),
    ('name="waypoint_03"', 'name="waypoint_09"', "consecutively"),
    ('<numeric name="flight_limits" data="3 2 3 12 0.7"/>', '<numeric name="flight_limits" data="3 2 3"/>', "flight_limits"),
    ('name="obstacle_02" type="box"', 'name="obstacle_02" type="sphere"', "axis-aligned box"),
    ('<site name="rotor_3" pos="-0.120208 0.120208 0"', '<site name="rotor_3" pos="-0.2 0.120208 0"', "symmetric"),
    ('gravity="0 0 -9.81"', 'gravity="0 0 0"', "gravity"),
])
def test_broken_scenes_raise_like_the_reference(tmp_path, old, new, msg):
    from uav_ac.simulation.mujoco_sim import MujocoSimulation
    text = open(SCENE).read()
    assert old in text
    p = tmp_path / "broken.xml"
    p.write_text(text.replace(old, new))
    with pytest.raises(ValueError, match=msg):
        MujocoSimulation(p)


def test_config_helpers_like_the_reference():
    """uav_ac.utils.get_config / parse_array (reference utils.py:8-28, config.ini:2,7,9)."""
    import configparser
    from uav_ac import utils
    cfg, flight = utils.get_config()
    assert cfg.getint("frequency") == 10
    assert flight.getfloat("velocity") == 3.0 and flight.getfloat("min_dist_target") == 0.5
    parser = configparser.ConfigParser()
    parser.read_string("[S]\nlimits = [[0, 0, 0], [10, 10, 10]]\n")
    assert np.array_equal(utils.parse_array(parser["S"], "limits"), np.array([[0, 0, 0], [10, 10, 10]]))


def test_default_scene_is_the_lab_course():
    """MujocoSimulation() without a path reads the packaged scene: same data as the test fixture."""
    from uav_ac.simulation.mujoco_sim import DEFAULT_SCENE_PATH, MujocoSimulation
    a, b = MujocoSimulation(), MujocoSimulation(SCENE)
    assert os.path.exists(DEFAULT_SCENE_PATH)
    assert np.array_equal(a.mission_waypoints, b.mission_waypoints) and np.array_equal(a.obstacles, b.obstacles)
    assert np.array_equal(a.space_limits, b.space_limits) and a.quad.m == b.quad.m
    assert a.has_collision is False and a.collision_detected is False
    a.quad.X[0:3] = [4.0, 7.0, -3.1]                        # the centre of obstacle_00
    assert a.has_collision is True
