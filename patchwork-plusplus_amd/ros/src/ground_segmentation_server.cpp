// ground_segmentation_server.cpp -- the rclcpp component "patchworkpp_node" around patchworkpp_ros::SegmentationCore.
// Same interface as the reference's node (ros/src/GroundSegmentationServer.cpp:24-73): parameters (segmentation_core.hpp),
// subscription "pointcloud_topic" with the sensor-data QoS, publishers "/patchworkpp/cloud", "/patchworkpp/ground",
// "/patchworkpp/nonground" (reliable, transient-local), ground / non-ground stamped with `base_frame`.
// NOT compiled in this repository's image (no ROS 2 there): the logic it calls is built and tested without ROS
// (examples/ros_core_demo.cpp, tests/test_gpu_parity.py::test_ros_node_core); this file is the thin part that needs rclcpp.
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <rclcpp/qos.hpp>
#include <rclcpp/rclcpp.hpp>
#include <rclcpp_components/register_node_macro.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <std_msgs/msg/header.hpp>

#include "patchworkpp_ros/segmentation_core.hpp"

namespace patchworkpp_ros {

class GroundSegmentationServer : public rclcpp::Node {
public:
    explicit GroundSegmentationServer(const rclcpp::NodeOptions &options) : rclcpp::Node("patchworkpp_node", options) {
        base_frame_ = declare_parameter<std::string>("base_frame", base_frame_);
        const patchwork::Params params = declare_parameters(
            [this](const std::string &name, double def) { return declare_parameter<double>(name, def); },
            [this](const std::string &name, int def) { return (int)declare_parameter<int>(name, def); },
            [this](const std::string &name, bool def) { return declare_parameter<bool>(name, def); });
        core_ = std::make_unique<SegmentationCore>(params, (int)declare_parameter<int>("device", 0));

        sub_ = create_subscription<sensor_msgs::msg::PointCloud2>(
            "pointcloud_topic", rclcpp::SensorDataQoS(), std::bind(&GroundSegmentationServer::on_cloud, this, std::placeholders::_1));
        rclcpp::QoS qos(rclcpp::QoSInitialization::from_rmw(rmw_qos_profile_default));
        qos.reliability(RMW_QOS_POLICY_RELIABILITY_RELIABLE);
        qos.durability(RMW_QOS_POLICY_DURABILITY_TRANSIENT_LOCAL);
        cloud_pub_ = create_publisher<sensor_msgs::msg::PointCloud2>("/patchworkpp/cloud", qos);
        ground_pub_ = create_publisher<sensor_msgs::msg::PointCloud2>("/patchworkpp/ground", qos);
        nonground_pub_ = create_publisher<sensor_msgs::msg::PointCloud2>("/patchworkpp/nonground", qos);
        RCLCPP_INFO(get_logger(), "Patchwork++ (MI355X) ROS 2 node initialized");
    }

private:
    void on_cloud(const sensor_msgs::msg::PointCloud2::ConstSharedPtr &msg) {
        std::vector<Field> fields;
        for (const auto &f : msg->fields) fields.push_back({f.name, f.offset, f.datatype, f.count});
        CloudView view;
        view.height = msg->height;
        view.width = msg->width;
        view.point_step = msg->point_step;
        view.fields = fields.data();
        view.num_fields = fields.size();
        view.data = msg->data.data();
        view.data_size = msg->data.size();
        SegmentationCore::Output out;
        try {  //  one malformed message must not take the node down: log it, drop it
            out = core_->estimate(view);
        } catch (const std::exception &e) {
            RCLCPP_ERROR(get_logger(), "message dropped: %s", e.what());
            return;
        }
        cloud_pub_->publish(to_msg(std::move(out.cloud), msg->header));
        std_msgs::msg::Header header = msg->header;
        header.frame_id = base_frame_;
        ground_pub_->publish(to_msg(std::move(out.ground), header));
        nonground_pub_->publish(to_msg(std::move(out.nonground), header));
    }
    static std::unique_ptr<sensor_msgs::msg::PointCloud2> to_msg(XyzCloud &&c, const std_msgs::msg::Header &header) {
        auto m = std::make_unique<sensor_msgs::msg::PointCloud2>();
        m->header = header;
        m->height = c.height;
        m->width = c.width;
        m->point_step = c.point_step;
        m->row_step = c.row_step;
        m->is_bigendian = false;
        m->is_dense = true;
        for (const Field &f : c.fields) {
            sensor_msgs::msg::PointField pf;
            pf.name = f.name;
            pf.offset = f.offset;
            pf.datatype = f.datatype;
            pf.count = f.count;
            m->fields.push_back(pf);
        }
        m->data = std::move(c.data);
        return m;
    }

    rclcpp::Subscription<sensor_msgs::msg::PointCloud2>::SharedPtr sub_;
    rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr cloud_pub_, ground_pub_, nonground_pub_;
    std::unique_ptr<SegmentationCore> core_;
    std::string base_frame_{"base_link"};
};

}  // namespace patchworkpp_ros

RCLCPP_COMPONENTS_REGISTER_NODE(patchworkpp_ros::GroundSegmentationServer)
