"""Command-line flags of train.py / test.py (same flag names as the reference's parse_args.py)."""


def add_args(parser):
    parser.add_argument("--config", type=str, default=None)
    parser.add_argument("--track", default=None, choices=["hand", "hand_IKNet", "obj_opt", False], help="tracking for test")
    parser.add_argument("--num_workers", type=int, default=0, help="num_workers in data_loader")
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--debug_save", action="store_true", default=False)
    parser.add_argument("--save", action="store_true", default=False)
    parser.add_argument("--data_config", type=str, default=None)
    parser.add_argument("--obj_category", type=str, default=None)
    parser.add_argument("--experiment_dir", type=str, default=None)
    parser.add_argument("--batch_size", type=int, default=None)
    parser.add_argument("--cuda_id", type=int, default=None)
    parser.add_argument("--total_epoch", default=None, type=int)
    parser.add_argument("--optimizer", type=str, default=None)
    parser.add_argument("--weight_decay", type=float, default=None)
    parser.add_argument("--learning_rate", type=float, default=None)
    parser.add_argument("--lr_policy", type=str, default=None)
    parser.add_argument("--lr_gamma", type=float, default=None)
    parser.add_argument("--lr_step_size", type=int, default=None)
    parser.add_argument("--lr_clip", type=float, default=None)
    parser.add_argument("--num_points", type=int, default=None)
    parser.add_argument("--freq/save", type=int, default=None, help="ckpt saving frequency in epochs")
    parser.add_argument("--pointnet_cfg/camera", type=str, default=None)
    parser.add_argument("--network/type", type=str, default=None)
    parser.add_argument("--network/backbone_out_dim", type=int, default=None)
    # additions of this repo (synthetic runs)
    parser.add_argument("--synthetic_frames", type=int, default=None, help="length of the synthetic train set / sequences")
    parser.add_argument("--max_iters", type=int, default=None, help="stop an epoch after this many iterations (smoke runs)")
    parser.add_argument("--hand_model", type=str, default=None, choices=["synthetic"],
                        help="track=hand_IKNet with use_optimization: the hand model of the pose optimiser ('synthetic' = the "
                             "linear-blend-skinning stand-in of models/hand_model.py; a MANO layer is passed programmatically)")
    parser.add_argument("--hand_particles", type=int, default=None, help="candidate hands per optimiser iteration (reference: 5120)")
    return parser
