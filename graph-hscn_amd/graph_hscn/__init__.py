"""graph_hscn -- MI355X-native hot path of Graph-HSCN (MinCUT coarsening +
heterogeneous local/virtual message passing) behind the reference's own import
paths: ``graph_hscn.model.hscn.{SCN,HSCN,build_hscn,build_conv_relation}``,
``graph_hscn.loader.hetero_data.{generate_hetero_data,hetero_loaders}``,
``graph_hscn.train.train_clustering.train_clustering``."""
__version__ = "0.1.0"
