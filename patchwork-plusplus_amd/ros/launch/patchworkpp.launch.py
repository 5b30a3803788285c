"""Launch the MI355X Patchwork++ node with the parameter set of the reference's launch file
(ros/launch/patchworkpp.launch.py:50-64 there; the same set is a parity variant of this repository's GPU tests:
tests/test_gpu_parity.py, ROS_LAUNCH)."""
from launch import LaunchDescription
from launch.actions import DeclareLaunchArgument
from launch.substitutions import LaunchConfiguration
from launch_ros.actions import Node

PATCHWORKPP = {
    "sensor_height": 1.88, "num_iter": 3, "num_lpr": 20, "num_min_pts": 0, "th_seeds": 0.3, "th_dist": 0.125,
    "th_seeds_v": 0.25, "th_dist_v": 0.9, "max_range": 80.0, "min_range": 1.0, "uprightness_thr": 0.101, "verbose": True,
}


def generate_launch_description():
    topic = LaunchConfiguration("cloud_topic", default="/pointcloud")
    base_frame = LaunchConfiguration("base_frame", default="base_link")
    use_sim_time = LaunchConfiguration("use_sim_time", default="true")
    node = Node(
        package="patchworkpp", executable="patchworkpp_node", name="patchworkpp_node", output="screen",
        remappings=[("pointcloud_topic", topic)],
        parameters=[dict(PATCHWORKPP, base_frame=base_frame, use_sim_time=use_sim_time)],
    )
    return LaunchDescription([DeclareLaunchArgument("cloud_topic", default_value="/pointcloud"), node])
