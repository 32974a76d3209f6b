"""Namespace with the option fields the registration plugin reads, at the reference's defaults
(options/base_options.py:26-70, options/train_options.py:13-41, registration_model.py:35-71 with
CUT_mode=CUT).  train.py users pass their own parsed `opt`; this is for tests / bench / smoke."""
import argparse


def default_options(**overrides):
    opt = argparse.Namespace(
        # base_options.py
        name='experiment_name', gpu_ids=[0], checkpoints_dir='./checkpoints', model='registration',
        input_nc=1, output_nc=1, ngf=64, netG='resnet_9blocks', normG='instance', init_type='xavier',
        init_gain=0.02, no_dropout=True, no_antialias=False, no_antialias_up=False, direction='AtoB',
        batch_size=1, load_size=256, crop_size=256, preprocess='resize_and_crop', epoch='latest', verbose=False,
        # train_options.py
        isTrain=True, continue_train=False, epoch_count=1, pretrained_name=None, n_epochs=150, n_epochs_decay=150,
        beta1=0.5, beta2=0.999, lr=0.0002, gan_mode='lsgan', pool_size=0, lr_policy='linear', lr_decay_iters=50,
        # registration_model.py (CUT defaults)
        CUT_mode='CUT', lambda_GAN=0.0, lambda_NCE=0.25, nce_idt=True, nce_layers='0,4,8,12,16',
        nce_includes_all_negatives_from_minibatch=False, netF='mlp_sample', netF_nc=256, nce_T=0.07,
        num_patches=256, flip_equivariance=False,
        dvf_image='synthetic',     # build-defined: None = ./deform256.jpg as in the reference (raises when absent)
        reuse_key_features=True,   # build-defined: tap NCE key features in forward() (exact, see registration_model.forward)
        batch_query_passes=True,   # build-defined: one encoder pass for the three NCE terms' query batches
        bucket_allreduce=False,    # build-defined (data-parallel): G's late layers start their all-reduce inside backward
        nce_sequential_keys=False,  # build-defined: key side of the NCE terms one netF call per term (reference order, host ids)
        skip_unused_target=True,   # build-defined: do not compute VxmDense's discarded warp(target, -flow) output
        global_mask_norm=False,    # build-defined, DDP: masked-L1 normalised by the GLOBAL batch's mask sums (DataParallel semantics)
        overlap_registration=True,  # build-defined: netR's forward / backward on a second HIP stream beside the generator's
        deterministic_wgrad=False,  # build-defined: weight / bias gradients by 64-bit fixed-point accumulation (bit-reproducible)
        staged_step=True,          # build-defined: the two-stream step as single-stream pieces (one linear hipGraph each when captured)
        capture_step=False)        # build-defined: replay the steady-state step as one hipGraph (REGISTRATIONModel)
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt
